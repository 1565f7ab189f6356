#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* __restrict__ coef, const double* __restrict__ x, double* __restrict__ out, int reps) {
    const int lane = threadIdx.x;
    double c = coef[lane & 15];          // lane k of every 16-lane row holds coefficient k
    double xv = x[lane];
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc0) : "v"(c), "v"(xv));
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc1) : "v"(c), "v"(xv));
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(c), "v"(xv));
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(acc3) : "v"(c), "v"(xv));
    }
    long t1 = clock64();
    out[lane] = acc0 + 10 * acc1 + 100 * acc2 + 1000 * acc3;
    if (lane == 0) out[64] = (double)(t1 - t0);
}
__global__ void k2(const double* __restrict__ coef, const double* __restrict__ x, double* __restrict__ out, int reps) {
    const int lane = threadIdx.x;
    double c = coef[lane & 15];
    double xv = x[lane];
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        asm("v_fmac_f64 %0, %1, %2" : "+v"(acc0) : "v"(c), "v"(xv));
        asm("v_fmac_f64 %0, %1, %2" : "+v"(acc1) : "v"(c), "v"(xv));
        asm("v_fmac_f64 %0, %1, %2" : "+v"(acc2) : "v"(c), "v"(xv));
        asm("v_fmac_f64 %0, %1, %2" : "+v"(acc3) : "v"(c), "v"(xv));
    }
    long t1 = clock64();
    out[lane] = acc0 + 10 * acc1 + 100 * acc2 + 1000 * acc3;
    if (lane == 0) out[64] = (double)(t1 - t0);
}
int main() {
    double hc[16], hx[64], ho[65];
    for (int i = 0; i < 16; i++) hc[i] = i + 1;
    for (int i = 0; i < 64; i++) hx[i] = 0.5 + i;
    double *c, *x, *o;
    hipMalloc(&c, sizeof(hc)); hipMalloc(&x, sizeof(hx)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(c, hc, sizeof(hc), hipMemcpyHostToDevice); hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int which = 0; which < 2; which++) {
        int reps = 1000;
        if (which == 0) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c, x, o, reps); else hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, c, x, o, reps);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        // expected (dpp): acc0 = reps*4*x, acc1 = reps*6*x, acc2 = reps*8*x, acc3 = reps*16*x
        double e = which == 0 ? reps * hx[17] * (4 + 60 + 800 + 16000) : reps * hx[17] * hc[1] * 1111;
        printf("%s lane17 %.1f expected %.1f  clock ticks %.0f per 4000 fmac\n", which == 0 ? "dpp" : "plain", ho[17], e, ho[64]);
    }
    return 0;
}
