// Microbenchmark: sustained fp64 rate of bare v_fma_f64 and v_mfma_f64_16x16x4_f64 loops on
// gfx950, to pin the `peak` the roofline fractions are quoted against (the on-disk microarch
// guide has no fp64 row; the datasheet says 78.6 TFLOP/s for both vector and matrix).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int ACC> __global__ void fma_loop(double* out, int iters, double a, double b) {
    double x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < ACC; i++) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC> __global__ void mfma_loop(double* out, int iters, double a, double b) {
    double4_t c[ACC];
#pragma unroll
    for (int i = 0; i < ACC; i++) c[i] = {0.0, 0.0, 0.0, 0.0};
    const double av = a + threadIdx.x * 1e-6, bv = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < ACC; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double* out;
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        fma_loop<8><<<blocks, threads>>>(out, iters, 0.999999, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * 8 * 8 * (double)iters * blocks * threads;
        printf("v_fma_f64   : %.3f ms  %.2f TFLOP/s\n", ms, flop / ms / 1e9);
        hipEventRecord(e0);
        mfma_loop<4><<<blocks, threads>>>(out, iters, 0.5, 1e-3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        flop = 2.0 * 16 * 16 * 4 * 4 * 4 * (double)iters * blocks * (threads / 64);
        printf("mfma_f64_16x16x4: %.3f ms  %.2f TFLOP/s\n", ms, flop / ms / 1e9);
    }
    return 0;
}
