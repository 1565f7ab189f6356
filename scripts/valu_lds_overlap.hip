// Development aid (r5): do fp64 vector instructions, matrix instructions and LDS traffic of ONE CU overlap, or add up?
// Stage A of p = 5 and p = 7 both measure "time = vector issue + matrix issue + LDS cycles" (profiles/r04_reg_kernel.txt, profiles/r03_mfma_n8.txt); this
// microbenchmark separates the hardware's part of that from the kernels' own (barriers, lockstep phases).  One 512-thread workgroup per CU (8 waves = 2 per
// SIMD, as the stage-A kernels run), every wave repeats a fixed instruction mix IT times; roles can differ between waves 0..3 and 4..7 (one of each per SIMD).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/valu_lds_overlap.hip -o scripts/bin/valu_lds_overlap      Run: scripts/bin/valu_lds_overlap [iterations = 4000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one group of the mix: F fp64 FMAs on eight independent accumulators, S ds_write_b64, L ds_read_b64, M v_mfma_f64_4x4x4_4b_f64; the LDS addresses are
// lane-linear (conflict-free); no wait inside a group
template <int K, int KN, class Fn> __device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (K < KN) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, KN>(f);
    }
}
template <int F, int S, int L, int M>
__device__ __forceinline__ void group(double (&a)[8], double (&m)[4], double b, double c, unsigned addr, double (&ld)[8]) {
    static_for<0, (F > S + L + M ? F : S + L + M)>([&a, &m, &ld, b, c, addr](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < F) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[k & 7]) : "v"(b), "v"(c));
        if constexpr (k < S) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(a[k & 7]), "n"((k & 15) * 512) : "memory");
        if constexpr (k < L) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ld[k & 7]) : "v"(addr), "n"((k & 15) * 512) : "memory");
        if constexpr (k < M) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[k & 3]) : "v"(b), "v"(c));
    });
}

// waves 0..3 run mix A, waves 4..7 mix B (0 instructions = the waves leave at once)
template <int FA, int SA, int LA, int MA, int FB, int SB, int LB, int MB>
__global__ void __launch_bounds__(512) mix_kernel(double* out, int iters, double b, double c) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    double a[8], m[4] = {0, 0, 0, 0}, ld[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = tid + k;
    const unsigned addr = (unsigned)((wave * 16 * 64 + (tid & 63)) * 8);          // each wave its own 8 KB: 16 slots of 512 B
    lds[tid] = 0.0;
    __syncthreads();
    if (wave < 4) {
        if (FA + SA + LA + MA > 0)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int g = 0; g < 4; g++) group<FA, SA, LA, MA>(a, m, b, c, addr, ld);
                if (SA + LA > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
    } else {
        if (FB + SB + LB + MB > 0)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int g = 0; g < 4; g++) group<FB, SB, LB, MB>(a, m, b, c, addr, ld);
                if (SB + LB > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += a[k] + ld[k];
    s += m[0] + m[1] + m[2] + m[3];
    if (s == 1.2345) out[blockIdx.x * 512 + tid] = s;
}

static double g_clock_ghz = 2.4;
template <int FA, int SA, int LA, int MA, int FB, int SB, int LB, int MB>
static int run(const char* what, double* out, int iters) {
    auto k = mix_kernel<FA, SA, LA, MA, FB, SB, LB, MB>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 64 * 1024, 0, out, iters, 1.0000001, 1e-9);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double cyc = best * 1e6 * g_clock_ghz / iters;                         // cycles per iteration (4 groups) at the nominal clock
    printf("%-86s %8.1f cycles per iteration\n", what, cyc);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    double* out; CK(hipMalloc(&out, 256 * 512 * 8));
    printf("one 512-thread workgroup per CU, 2 waves per SIMD; an iteration = 4 groups; per CU and iteration, at 2.4 GHz:\n");
    // --- all eight waves the same mix
    run<16, 0, 0, 0, 16, 0, 0, 0>("all waves: 64 FMA                          (issue bound: 2 waves x 64 x 4 = 512)", out, iters);
    run<0, 4, 0, 0, 0, 4, 0, 0>("all waves: 16 ds_write_b64                 (table: 128 x 6 = 768)", out, iters);
    run<0, 0, 8, 0, 0, 0, 8, 0>("all waves: 32 ds_read_b64                  (table: 256 x 2 = 512)", out, iters);
    run<0, 0, 0, 4, 0, 0, 0, 4>("all waves: 16 MFMA 4x4x4                   (2 waves x 16 x 16 = 512)", out, iters);
    run<16, 4, 0, 0, 16, 4, 0, 0>("all waves: 64 FMA + 16 stores interleaved  (max 768, sum 1280)", out, iters);
    run<16, 0, 8, 0, 16, 0, 8, 0>("all waves: 64 FMA + 32 loads interleaved   (max 512, sum 1024)", out, iters);
    run<16, 4, 8, 0, 16, 4, 8, 0>("all waves: 64 FMA + 16 stores + 32 loads   (max 1280 LDS, sum 1792)", out, iters);
    run<16, 0, 0, 4, 16, 0, 0, 4>("all waves: 64 FMA + 16 MFMA interleaved    (max 512, sum 1024)", out, iters);
    run<0, 4, 0, 4, 0, 4, 0, 4>("all waves: 16 stores + 16 MFMA             (max 768, sum 1280)", out, iters);
    run<16, 4, 8, 4, 16, 4, 8, 4>("all waves: 64 FMA + 16 st + 32 ld + 16 MFMA (sum 2304)", out, iters);
    // --- roles: waves 0..3 arithmetic, waves 4..7 LDS
    run<32, 0, 0, 0, 0, 0, 0, 0>("waves 0-3: 128 FMA, waves 4-7 idle         (1 wave x 128 x 4 = 512)", out, iters);
    run<0, 0, 0, 0, 0, 8, 0, 0>("waves 0-3 idle, waves 4-7: 32 stores       (128 x 6 = 768)", out, iters);
    run<32, 0, 0, 0, 0, 8, 0, 0>("waves 0-3: 128 FMA | waves 4-7: 32 stores  (max 768, sum 1280)", out, iters);
    run<0, 0, 0, 0, 0, 0, 16, 0>("waves 0-3 idle, waves 4-7: 64 loads        (256 x 2 = 512)", out, iters);
    run<32, 0, 0, 0, 0, 0, 16, 0>("waves 0-3: 128 FMA | waves 4-7: 64 loads   (max 512, sum 1024)", out, iters);
    run<32, 0, 0, 0, 0, 8, 16, 0>("waves 0-3: 128 FMA | waves 4-7: 32 st + 64 ld (max 1280, sum 1792)", out, iters);
    run<0, 0, 0, 8, 0, 0, 0, 0>("waves 0-3: 32 MFMA, waves 4-7 idle         (32 x 16 = 512)", out, iters);
    run<0, 0, 0, 8, 32, 0, 0, 0>("waves 0-3: 32 MFMA | waves 4-7: 128 FMA    (max 512, sum 1024)", out, iters);
    run<0, 0, 0, 8, 0, 8, 0, 0>("waves 0-3: 32 MFMA | waves 4-7: 32 stores  (max 768, sum 1280)", out, iters);
    run<32, 0, 0, 8, 0, 8, 16, 0>("waves 0-3: 128 FMA + 32 MFMA | waves 4-7: 32 st + 64 ld", out, iters);
    return 0;
}
