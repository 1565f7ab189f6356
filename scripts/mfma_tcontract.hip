// Diagnostic: the time contraction of the Picard iteration (stage A, 3-D, N = 6, 5 variables)
//     q[l'][x] = u[x] - dt * sum_l T[l'][l] * (S_x + S_y + S_z)[l][x]
// on LDS-resident data, once with the vector ALU (the product kernel's form) and once on the matrix pipe
// (v_mfma_f64_16x16x4_f64: rows = output levels (6 of 16 used), k = input levels (6 of 8), columns = 16 nodes of one
// variable), alone and beside waves that do flux-like vector work -- does the matrix pipe take the contraction off the VALU?
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/mfma_tcontract.hip -o scripts/bin/mfma_tcontract
// Run  : scripts/bin/mfma_tcontract          (prints ms per pass and per-pass ratios; counters: scripts/pmc_mfma.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 6, NV = 5, NN = 216, SL = 216, NTS = 6;
constexpr int ASZ = NV * NTS * SL;              // one q-sized array (doubles)
constexpr int NT = 768;                         // 12 waves, one workgroup per CU (the product kernel's shape)
constexpr size_t LDS_BYTES = (size_t)3 * ASZ * sizeof(double);
typedef double d4 __attribute__((ext_vector_type(4)));

__constant__ double c_T[36];                    // -dt * T

// ---- vector-ALU form: thread = (node, variable pair) as in the product kernel (3 groups x 256 lanes, 216 active)
__device__ inline void t_valu(double* lds, int tid, const double* u) {
    const int grp = tid >> 8, bt = tid & 255;
    if (bt >= NN) return;
    const int v0 = grp == 0 ? 0 : (grp == 1 ? 1 : 3), cnt = grp == 0 ? 1 : 2;
    for (int vv = 0; vv < cnt; vv++) {
        const int v = v0 + vv;
        double S[N];
#pragma unroll
        for (int l = 0; l < N; l++) {
            const int o = (v * NTS + l) * SL + bt;
            S[l] = lds[o] + lds[o + ASZ] + lds[o + 2 * ASZ];
        }
        const double uv = u[v];
#pragma unroll
        for (int lp = 0; lp < N; lp++) {
            double acc = uv;
#pragma unroll
            for (int l = 0; l < N; l++) acc += c_T[lp * N + l] * S[l];
            lds[(v * NTS + lp) * SL + bt] = acc;
        }
    }
}

// ---- matrix-pipe form: one wave per tile of 16 nodes of one variable; 70 tiles per cell (14 per variable, the last half empty)
__device__ inline void t_mfma(double* lds, int tid, const double* u) {
    const int wave = tid >> 6, lane = tid & 63;
    const int k = lane >> 4, j = lane & 15;
    // A operand: lane (i = lane & 15, k = lane >> 4) holds A[i][k]; rows >= 6 are zero
    const int i = lane & 15;
    const double a0 = i < N ? c_T[i * N + k] : 0.0;
    const double a1 = (i < N && k < 2) ? c_T[i * N + 4 + k] : 0.0;
    for (int tile = wave; tile < 70; tile += NT / 64) {
        const int v = tile / 14, c = tile - v * 14;
        const int node = c * 16 + j;
        const bool ok = node < NN;
        const int o0 = (v * NTS + k) * SL + (ok ? node : 0);
        const int o1 = (v * NTS + 4 + (k < 2 ? k : 0)) * SL + (ok ? node : 0);
        double b0 = lds[o0] + lds[o0 + ASZ] + lds[o0 + 2 * ASZ];
        double b1 = lds[o1] + lds[o1 + ASZ] + lds[o1 + 2 * ASZ];
        if (k >= 2) b1 = 0.0;
        d4 acc = {u[v], u[v], u[v], u[v]};                      // C: every output level starts from u
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
        // D: lane (col = lane & 15, row = (lane >> 4) + 4 * reg)
        if (ok) {
            lds[(v * NTS + k) * SL + node] = acc[0];
            if (k < 2) lds[(v * NTS + 4 + k) * SL + node] = acc[1];
        }
    }
}

// ---- flux-like vector work without LDS: NF dependent-chain groups of fp64 FMAs per call (stands for the derivative phase)
__device__ inline double valu_work(double x, int reps) {
    double a = x, b = x + 1.0, c = x + 2.0, d = x + 3.0;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            a = fma(a, 1.0000001, 1e-9);
            b = fma(b, 0.9999999, 1e-9);
            c = fma(c, 1.0000002, -1e-9);
            d = fma(d, 0.9999998, -1e-9);
        }
    }
    return a + b + c + d;
}

// mode 0: VALU contraction, all waves.  1: MFMA contraction, all waves.  2: flux-like VALU work only.
// 3: flux-like work on every wave + VALU contraction.  4: flux-like work on every wave + MFMA contraction.
__global__ void __launch_bounds__(NT) bench_kernel(int mode, int passes, int reps, double* out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * ASZ; i += NT) lds[i] = 1e-3 * ((i * 2654435761u) % 1000) / 1000.0;
    __syncthreads();
    double u[NV] = {1.0, 0.1, 0.2, 0.3, 2.5};
    double w = 0.0;
    for (int p = 0; p < passes; p++) {
        if (mode == 2 || mode == 3 || mode == 4) w += valu_work(u[0] + p, reps);
        if (mode == 0 || mode == 3) t_valu(lds, tid, u);
        if (mode == 1 || mode == 4) t_mfma(lds, tid, u);
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = lds[5] + w;
}

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 2000;
    const int reps = argc > 2 ? atoi(argv[2]) : 12;            // 12 x 32 FMAs per lane and pass ~ the flux + D work per T phase
    double T[36];
    for (int i = 0; i < 36; i++) T[i] = -1e-3 * (0.1 + 0.01 * i);
    hipMemcpyToSymbol(HIP_SYMBOL(c_T), T, sizeof(T));
    double* out;
    hipMalloc(&out, 256 * sizeof(double));
    hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    // check: both forms give the same q
    {
        // (a pass of mode 0 and of mode 1 from the same data: compared through the checksum of one workgroup)
        double r[2];
        for (int m = 0; m < 2; m++) {
            hipLaunchKernelGGL(bench_kernel, dim3(1), dim3(NT), LDS_BYTES, 0, m, 1, 0, out);
            hipMemcpy(&r[m], out, sizeof(double), hipMemcpyDeviceToHost);
        }
        printf("check  q[5] after one pass: valu %.15g  mfma %.15g  (diff %.2e)\n", r[0], r[1], r[0] - r[1]);
    }
    const char* names[5] = {"VALU contraction", "MFMA contraction", "flux-like VALU work alone", "flux-like work + VALU contraction",
                            "flux-like work + MFMA contraction"};
    float ms[5];
    for (int round = 0; round < 2; round++)
        for (int m = 0; m < 5; m++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(bench_kernel, dim3(256), dim3(NT), LDS_BYTES, 0, m, passes, reps, out);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms[m], e0, e1);
            if (round == 1) printf("mode %d  %-36s %8.3f ms  = %7.1f ns per pass and CU\n", m, names[m], ms[m], ms[m] * 1e6 / passes);
        }
    printf("contraction alone      : MFMA / VALU = %.2f\n", ms[1] / ms[0]);
    printf("beside flux-like work  : (work + MFMA) / (work + VALU) = %.3f   work alone %.3f ms, + VALU %.3f, + MFMA %.3f\n",
           ms[4] / ms[3], ms[2], ms[3], ms[4]);
    return 0;
}
