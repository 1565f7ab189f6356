// One instantiation unit of the ADER-DG kernels, compiled once per
// (EXA_DIM, EXA_PDE_ID) pair (see exahype_amd/build.py) so the heavy fully
// unrolled kernels build in parallel.
#include <cstdio>
#include "exa_dg_kernels.hpp"
#include "exa_launch.hpp"

#ifndef EXA_DIM
#error "compile with -DEXA_DIM=2|3 -DEXA_PDE_ID=0|1|2"
#endif

namespace exa {

#if EXA_PDE_ID == 0
using PDE = EulerRef2D;
#elif EXA_PDE_ID == 1
using PDE = Euler;
#elif EXA_PDE_ID == 2
using PDE = Advection<1>;
#endif

constexpr int DIM = EXA_DIM;
constexpr int SCRATCH_BLOCKS = 256;            // persistent grid of the scratch variant: one workgroup per CU
#if EXA_DIM == 3
constexpr int MAX_N = 8;                       // N = 7, 8: cell image > 160 KiB of LDS -> scratch variant (functional fallback)
constexpr int NT_A = 256;                      // base threads (tasks per phase ~ CPB * N^3 <= 256)
__host__ __device__ constexpr int cpb_of(int N) { return N == 2 ? 16 : N == 3 ? 8 : N == 4 ? 4 : N == 5 ? 2 : 1; }
#ifndef EXA_HS
#define EXA_HS 1
#endif
__host__ __device__ constexpr int hs_of(int N) { return (EXA_HS == 2 && N % 2 == 0) ? 2 : 1; }   // 2: row split -> 512 threads, 2 waves/SIMD
#else
constexpr int MAX_N = 8;
constexpr int NT_A = 128;
__host__ __device__ constexpr int cpb_of(int N) { return (128 / (N * N)) > 0 ? 128 / (N * N) : 1; }
__host__ __device__ constexpr int hs_of(int) { return 1; }
#endif

template <int N> static DgOps<N> pack_ops(const DgOpsHost* h) {
    DgOps<N> o;
    for (int i = 0; i < N; i++) {
        o.w[i] = h->w[i];
        o.iw[i] = 1.0 / h->w[i];
        o.phiL[i] = h->phiL[i];
        o.phiR[i] = h->phiR[i];
        o.Tsum[i] = 0.0;
        for (int j = 0; j < N; j++) o.Tsum[i] += h->iK1[i * N + j] * h->w[j];
        for (int j = 0; j < N; j++) {
            o.D[i * N + j] = h->D[i * N + j];
            o.DT[j * N + i] = h->D[i * N + j];
            o.Kxi[i * N + j] = h->Kxi[i * N + j];
            o.T[i * N + j] = h->iK1[i * N + j] * h->w[j];
        }
    }
    return o;
}

// device image of the operator block for this N (uploaded once per plan by capi.cpp)
template <int N> static void fill_ops(const DgOpsHost* h, void* dst) { *static_cast<DgOps<N>*>(dst) = pack_ops<N>(h); }

template <int N> constexpr bool needs_scratch() { return StageA<DIM, N, PDE, cpb_of(N)>::LDS_BYTES > 160 * 1024; }

template <int N>
static int launch_a(const double* u_in, double* u_out, double* trace, long ncells, const CellBox* box, double dt,
                    const double* idx, int n_it, const DgOpsHost* ops, hipStream_t s) {
    constexpr int CPB = cpb_of(N);
    if (box->nbox <= 0) return 0;
    if constexpr (needs_scratch<N>()) {
        if (!ops->scratch) { set_error("stage_a: scratch slab missing for N = %d", N); return -1; }
        hipLaunchKernelGGL((dg_stage_a_scratch_kernel<DIM, N, PDE, 1, 256>), dim3(SCRATCH_BLOCKS), dim3(256), 0, s, u_in, u_out,
                           trace, ncells, *box, dt, idx[0], idx[1], idx[2], n_it, pack_ops<N>(ops), static_cast<double*>(ops->scratch));
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) { set_error("stage_a (scratch) launch (dim %d, N %d): %s", DIM, N, hipGetErrorString(e0)); return -2; }
        return 0;
    } else {
    using SA = StageA<DIM, N, PDE, CPB>;
    auto kern = dg_stage_a_kernel<DIM, N, PDE, CPB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)SA::LDS_BYTES);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(stage_a, %zu B LDS): %s", SA::LDS_BYTES, hipGetErrorString(e));
            return -2;
        }
        attr_set = true;
    }
    if (n_it == 0) {            // single-stage variant: lean LDS image, many cells per workgroup
        constexpr int NT1 = 256;
        constexpr int NNc = ipow(N, DIM);
        constexpr int CPB1 = (NT1 / NNc) > 0 ? NT1 / NNc : 1;
        const long nb1 = (box->nbox + CPB1 - 1) / CPB1;
        hipLaunchKernelGGL((dg_stage_a_single_kernel<DIM, N, PDE, CPB1, NT1>), dim3((unsigned)nb1), dim3(NT1), 0, s, u_in, u_out,
                           trace, ncells, *box, dt, idx[0], idx[1], idx[2], ops->dev);
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) {
            set_error("stage_a (single stage) launch (dim %d, N %d): %s", DIM, N, hipGetErrorString(e1));
            return -2;
        }
        return 0;
    }
    const long nblocks = (box->nbox + CPB - 1) / CPB;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(SA::NT), SA::LDS_BYTES, s, u_in, u_out, trace, ncells, *box, dt,
                       idx[0], idx[1], idx[2], n_it, ops->dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("stage_a launch (dim %d, N %d): %s", DIM, N, hipGetErrorString(e));
        return -2;
    }
    return 0;
    }
}

template <int N> static size_t scratch_bytes_n() {
    if constexpr (needs_scratch<N>()) return (size_t)SCRATCH_BLOCKS * StageAScratch<DIM, N, PDE, 1>::LDS_BYTES;
    else return 0;
}

template <int N>
static int launch_b(double* u, const double* trace, const StageBBox* box, long ncells, double dt, const double* idx,
                    const DgOpsHost* ops, hipStream_t s) {
    constexpr int NFc = ipow(N, DIM - 1);
    constexpr bool SHUFFLE = pow2ceil(NFc) == NFc;                    // face fills its lane segment
    // shuffle variant: one lane segment per face, all faces of the workgroup's cells in one pass;
    // dense variant: (cell, face, node) tasks packed over 256 lanes
    constexpr int PER = 2 * DIM * (SHUFFLE ? pow2ceil(NFc) : NFc);
    constexpr int CPB = PER >= 256 ? 1 : 256 / PER;
    constexpr int NT_B = SHUFFLE ? ((CPB * PER + 63) / 64) * 64 : 256;
    StageBArgs A;
    long nbox = 1;
    for (int d = 0; d < 3; d++) {
        A.nc[d] = box->nc[d];
        A.lo[d] = box->lo[d];
        A.nb[d] = box->nb[d];
        nbox *= box->nb[d];
    }
    for (int f = 0; f < 6; f++) A.ghost[f] = box->ghost[f];
    if (nbox <= 0) return 0;
    const long nblocks = (nbox + CPB - 1) / CPB;
    if constexpr (SHUFFLE)
        hipLaunchKernelGGL((dg_stage_b_kernel<DIM, N, PDE, CPB, NT_B>), dim3((unsigned)nblocks), dim3(NT_B), 0, s, u, trace, A,
                           ncells, nbox, dt, idx[0], idx[1], idx[2], pack_ops<N>(ops));
    else
        hipLaunchKernelGGL((dg_stage_b_dense_kernel<DIM, N, PDE, CPB, NT_B>), dim3((unsigned)nblocks), dim3(NT_B), 0, s, u, trace,
                           A, ncells, nbox, dt, idx[0], idx[1], idx[2], pack_ops<N>(ops));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("stage_b launch (dim %d, N %d): %s", DIM, N, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

#define EXA_N_CASES(X) X(2) X(3) X(4) X(5) X(6)
#define EXA_N_CASES_HI(X) X(7) X(8)

static int stage_a(int N, const double* u_in, double* u_out, double* trace, long ncells, const CellBox* box, double dt,
                   const double* idx, int n_it, const DgOpsHost* ops, hipStream_t s) {
    switch (N) {
#define X(n) case n: return launch_a<n>(u_in, u_out, trace, ncells, box, dt, idx, n_it, ops, s);
        EXA_N_CASES(X)
        EXA_N_CASES_HI(X)
#undef X
    }
    set_error("ADER-DG stage A: N = %d is not built for dim %d (supported 2..%d)", N, DIM, MAX_N);
    return -1;
}

static int stage_b(int N, double* u, const double* trace, const StageBBox* box, long ncells, double dt,
                   const double* idx, const DgOpsHost* ops, hipStream_t s) {
    switch (N) {
#define X(n) case n: return launch_b<n>(u, trace, box, ncells, dt, idx, ops, s);
        EXA_N_CASES(X)
        EXA_N_CASES_HI(X)
#undef X
    }
    set_error("ADER-DG stage B: N = %d is not built for dim %d (supported 2..%d)", N, DIM, MAX_N);
    return -1;
}

static size_t ops_image(int N, const DgOpsHost* h, void* dst) {
    switch (N) {
#define X(n) case n: if (dst) fill_ops<n>(h, dst); return sizeof(DgOps<n>);
        EXA_N_CASES(X)
        EXA_N_CASES_HI(X)
#undef X
    }
    return 0;
}

static size_t scratch_bytes(int N) {
    switch (N) {
#define X(n) case n: return scratch_bytes_n<n>();
        EXA_N_CASES(X)
        EXA_N_CASES_HI(X)
#undef X
    }
    return 0;
}

static int maxeig(const double* u, long nnodes, double* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), s);
    if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return -2; }
    long nb = (nnodes + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL((dg_maxeig_kernel<DIM, PDE>), dim3((unsigned)nb), dim3(256), 0, s, u, nnodes, out);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("maxeig launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

#define EXA_CAT_(a, b, c) a##b##_##c
#define EXA_CAT(a, b, c) EXA_CAT_(a, b, c)
// (a function, not a const global: a const global with a constant initialiser is also emitted for the device)
const DgLaunchTable* EXA_CAT(dg_table_, EXA_DIM, EXA_PDE_ID)() {
    static DgLaunchTable t;
    t.nv = PDE::NV;
    t.max_n = MAX_N;
    t.stage_a = stage_a;
    t.stage_b = stage_b;
    t.maxeig = maxeig;
    t.ops_image = ops_image;
    t.scratch_bytes = scratch_bytes;
    return &t;
}

}  // namespace exa
