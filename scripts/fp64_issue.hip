// fp64 FMA throughput per CU as a function of waves per SIMD (1 workgroup per CU):
// does a lone wave on a SIMD reach the 4-cycle v_fma_f64 pipe rate?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ACC> __global__ void fma_loop(double* out, int iters, double a, double b) {
    double x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < ACC; i++) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 8192;
    for (int threads : {256, 512, 768, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            fma_loop<16><<<256, threads>>>(out, iters, 0.999999, 1e-9);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = 2.0 * 16 * 4 * (double)iters * 256 * threads;
            if (rep) printf("1 WG/CU x %4d threads (%d waves/SIMD), 16 independent chains: %.3f ms  %.2f TFLOP/s  -> %.2f cycles per wave-FMA @2.4GHz\n",
                            threads, threads / 256, ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (16.0 * 4 * iters));
        }
    }
    return 0;
}
