// FV subcell limiter glue for the ADER-DG path (BASELINE configs[4]; no counterpart in the
// reference).  Two small kernels around the fused FV Rusanov patch kernel (fv_rusanov.hip, the reference's kernel
// shape with patch_size = 2p+1, halo_size = 1):
//   limiter_project_kernel      troubled cell -> FV patch: interior = (P x P x P) u, face halos = the adjacent
//                               subcell layer of the face neighbours' projections (periodic in the block), edge
//                               and corner halo entries = nearest interior value (never read by the 7-point stencil)
//   limiter_reconstruct_kernel  FV patch interior -> DG nodes: u = (R x R x R) v
// One workgroup per troubled cell.  Five-variable systems take the all-variables kernels further down (AoS runs, 28 barriers per cell);
// the kernels here (other variable counts) run the tensor products variable by variable through two LDS buffers.  The kernels are
// instantiated per order (N, N_s compile-time): with run-time extents every output element paid three integer divisions
// (~100 instructions) for its 8 FMAs -- 7.6 ms per step for the 13 K patches of cfg 4's per-GPU shape, 8 % of the step.
#include <cstdio>
#include <cstdlib>
#include "exa_launch.hpp"

namespace exa {

constexpr int LIM_MAX = 15 * 15 * 15;      // largest intermediate: Ns^3 at p = 7

// out[.., r, ..] = sum_c M[r0 + r][c] in[.., c, ..] along `axis` of a row-major array with extents e[0..dim)
// (extent C along `axis` on input, nr on output)
// C: compile-time contraction length (N or N_s); the extents along the other axes take few distinct values, so the
// divisions by `inner` / `nr` are done once per element with 32-bit unsigned arithmetic on small numbers
template <int C>
__device__ inline void lim_apply(const double* __restrict__ M, int ldm, int r0, int nr, const double* in, double* out,
                                 int dim, const int* e, int axis) {
    unsigned inner = 1, outer = 1;
    for (int a = axis + 1; a < dim; a++) inner *= e[a];
    for (int a = 0; a < axis; a++) outer *= e[a];
    const unsigned total = outer * nr * inner;
    for (unsigned t = threadIdx.x; t < total; t += blockDim.x) {
        const unsigned q = t / inner, in_i = t - q * inner;
        const unsigned o = q / (unsigned)nr, r = q - o * nr;
        const double* mr = M + (r0 + r) * ldm;
        const double* ip = in + (o * C) * inner + in_i;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < C; c++) acc += mr[c] * ip[c * inner];
        out[t] = acc;
    }
}

template <int DIM, int N>
__global__ void __launch_bounds__(256)
limiter_project_kernel(int, int, int nv, long nc0, long nc1, long nc2, const double* __restrict__ u,
                       const long* __restrict__ cells, double* __restrict__ patch, const double* __restrict__ P,
                       LimGhosts gh) {
    __shared__ double A[LIM_MAX], B[LIM_MAX];
    constexpr int Ns = 2 * N - 1;
    const long cell = cells[blockIdx.x];
    if (cell < 0) return;                                          // empty slot of a capacity-sized cell list
    const int S = Ns + 2;
    long NN = 1, SS = 1, NsD = 1;
    for (int a = 0; a < DIM; a++) { NN *= N; SS *= S; NsD *= Ns; }
    double* pt = patch + (long)blockIdx.x * SS * nv;
    const long nc[3] = {nc0, nc1, nc2};
    long cc[3];
    { long b = cell; cc[2] = DIM == 3 ? b % nc2 : 0; if (DIM == 3) b /= nc2; cc[1] = b % nc1; cc[0] = b / nc1; }
    for (int v = 0; v < nv; v++) {
        // ---- own projection
        for (int t = threadIdx.x; t < NN; t += blockDim.x) A[t] = u[(cell * NN + t) * nv + v];
        __syncthreads();
        int e[3] = {N, N, N};
        double *src = A, *dst = B;
        for (int a = 0; a < DIM; a++) {
            lim_apply<N>(P, N, 0, Ns, src, dst, DIM, e, a);
            e[a] = Ns;
            __syncthreads();
            double* tmp = src; src = dst; dst = tmp;
        }
        // patch entry (I,J,K): nearest interior value; the face layers are overwritten below
        for (int t = threadIdx.x; t < SS; t += blockDim.x) {
            int I[3], r = t;
            for (int a = DIM - 1; a >= 0; a--) { I[a] = r % S; r /= S; }
            int flat = 0;
            for (int a = 0; a < DIM; a++) {
                int ci = I[a] - 1;
                ci = ci < 0 ? 0 : (ci > Ns - 1 ? Ns - 1 : ci);
                flat = flat * Ns + ci;
            }
            pt[(long)t * nv + v] = src[flat];
        }
        __syncthreads();
        // ---- face halos from the neighbours' projections (one subcell layer each)
        for (int a = 0; a < DIM; a++)
            for (int side = 0; side < 2; side++) {
                const long layer = NsD / Ns;
                const double* g = gh.layer[a * 2 + side];
                if (g && cc[a] == (side ? nc[a] - 1 : 0)) {
                    // block boundary of a sharded grid: the neighbour's adjacent subcell layer arrived by exchange,
                    // layout [transverse cell][transverse subcell][var] (limiter_face_layers_kernel on the neighbour)
                    long tc = 0;
                    for (int b = 0; b < DIM; b++)
                        if (b != a) tc = tc * nc[b] + cc[b];
                    const double* gl = g + tc * layer * nv;
                    for (int t = threadIdx.x; t < layer; t += blockDim.x) {
                        int r = t, idx[3] = {0, 0, 0};
                        for (int b = DIM - 1; b >= 0; b--) {
                            if (b == a) continue;
                            idx[b] = r % Ns + 1;
                            r /= Ns;
                        }
                        idx[a] = side ? S - 1 : 0;
                        long flat = 0;
                        for (int b = 0; b < DIM; b++) flat = flat * S + idx[b];
                        pt[flat * nv + v] = gl[(long)t * nv + v];
                    }
                    __syncthreads();
                    continue;
                }
                long nb[3] = {cc[0], cc[1], cc[2]};
                nb[a] = (nb[a] + (side ? 1 : nc[a] - 1)) % nc[a];
                const long ncell = (nb[0] * nc1 + nb[1]) * (DIM == 3 ? nc2 : 1) + nb[2];
                for (int t = threadIdx.x; t < NN; t += blockDim.x) A[t] = u[(ncell * NN + t) * nv + v];
                __syncthreads();
                int e2[3] = {N, N, N};
                double *s2 = A, *d2 = B;
                for (int bb = 0; bb < DIM; bb++) {
                    const int b = (a + bb) % DIM;                // normal axis first: the other passes see one layer only
                    if (b == a) lim_apply<N>(P, N, side ? 0 : Ns - 1, 1, s2, d2, DIM, e2, b);     // the adjacent layer only
                    else lim_apply<N>(P, N, 0, Ns, s2, d2, DIM, e2, b);
                    e2[b] = (b == a) ? 1 : Ns;
                    __syncthreads();
                    double* tmp = s2; s2 = d2; d2 = tmp;
                }
                for (int t = threadIdx.x; t < layer; t += blockDim.x) {
                    // t enumerates the transverse subcells (axes != a, lexicographic)
                    int r = t, idx[3] = {0, 0, 0};
                    for (int b = DIM - 1; b >= 0; b--) {
                        if (b == a) continue;
                        idx[b] = r % Ns + 1;
                        r /= Ns;
                    }
                    idx[a] = side ? S - 1 : 0;
                    long flat = 0;
                    for (int b = 0; b < DIM; b++) flat = flat * S + idx[b];
                    pt[flat * nv + v] = s2[t];
                }
                __syncthreads();
            }
    }
}

// Sharded grids: the subcell layer next to the block face (a, side) of every cell of the block's boundary layer whose
// neighbour across the face is troubled (need[t] != 0; need == nullptr: all) -- what that neighbour's patch needs as
// halo.  out[transverse cell][transverse subcell][var]; one workgroup per transverse cell.
template <int DIM, int N>
__global__ void __launch_bounds__(256)
limiter_face_layers_kernel(int, int, int nv, long nc0, long nc1, long nc2, const double* __restrict__ u, int a, int side,
                           const double* __restrict__ need, double* __restrict__ out, const double* __restrict__ P) {
    __shared__ double A[LIM_MAX], B[LIM_MAX];
    constexpr int Ns = 2 * N - 1;
    const long tc = blockIdx.x;
    if (need && need[tc] == 0.0) return;
    const long nc[3] = {nc0, nc1, nc2};
    long NN = 1, NsD = 1;
    for (int b = 0; b < DIM; b++) { NN *= N; NsD *= Ns; }
    const long layer = NsD / Ns;
    long cc[3] = {0, 0, 0};
    { long r = tc; for (int b = DIM - 1; b >= 0; b--) { if (b == a) continue; cc[b] = r % nc[b]; r /= nc[b]; } }
    cc[a] = side ? nc[a] - 1 : 0;
    const long cell = (cc[0] * nc1 + cc[1]) * (DIM == 3 ? nc2 : 1) + cc[2];
    for (int v = 0; v < nv; v++) {
        for (int t = threadIdx.x; t < NN; t += blockDim.x) A[t] = u[(cell * NN + t) * nv + v];
        __syncthreads();
        int e[3] = {N, N, N};
        double *src = A, *dst = B;
        for (int bb = 0; bb < DIM; bb++) {
            const int b = (a + bb) % DIM;                        // normal axis first
            if (b == a) lim_apply<N>(P, N, side ? Ns - 1 : 0, 1, src, dst, DIM, e, b);     // the layer at the face only
            else lim_apply<N>(P, N, 0, Ns, src, dst, DIM, e, b);
            e[b] = (b == a) ? 1 : Ns;
            __syncthreads();
            double* tmp = src; src = dst; dst = tmp;
        }
        for (int t = threadIdx.x; t < layer; t += blockDim.x) out[(tc * layer + t) * nv + v] = src[t];
        __syncthreads();
    }
}

template <int DIM, int N>
__global__ void __launch_bounds__(256)
limiter_reconstruct_kernel(int, int, int nv, const double* __restrict__ patch, const long* __restrict__ cells,
                           double* __restrict__ u, const double* __restrict__ R) {
    __shared__ double A[LIM_MAX], B[LIM_MAX];
    constexpr int Ns = 2 * N - 1;
    const long cell = cells[blockIdx.x];
    if (cell < 0) return;                                          // empty slot
    const int S = Ns + 2;
    long NN = 1, SS = 1, NsD = 1;
    for (int a = 0; a < DIM; a++) { NN *= N; SS *= S; NsD *= Ns; }
    const double* pt = patch + (long)blockIdx.x * SS * nv;
    for (int v = 0; v < nv; v++) {
        for (int t = threadIdx.x; t < NsD; t += blockDim.x) {
            int r = t;
            long flat = 0, mul = 1;
            for (int a = DIM - 1; a >= 0; a--) { flat += (long)(r % Ns + 1) * mul; mul *= S; r /= Ns; }
            A[t] = pt[flat * nv + v];
        }
        __syncthreads();
        int e[3] = {Ns, Ns, Ns};
        double *src = A, *dst = B;
        for (int a = 0; a < DIM; a++) {
            lim_apply<Ns>(R, Ns, 0, N, src, dst, DIM, e, a);
            e[a] = N;
            __syncthreads();
            double* tmp = src; src = dst; dst = tmp;
        }
        for (int t = threadIdx.x; t < NN; t += blockDim.x) u[(cell * NN + t) * nv + v] = src[t];
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------------------------
// All variables at once (r2).  The kernels above run the tensor products variable by variable: every global access is an
// 8-byte element at a stride of nv doubles (five passes over the same lines), and a cell costs ~170 workgroup barriers (per
// variable: own projection 4, six face layers 5 each).  Here a lane carries all NV variables of an output element: the
// cell's u, the patch and the face layers move as contiguous AoS runs, the last pass of every product writes straight to
// global memory, and a cell costs 28 barriers.  LDS: two buffers, max(N^3, N_s^2 N) and N_s N^2 elements of NV doubles
// (110 KB at N = 8, NV = 5: one workgroup per CU).  Same summation order as above: bit-identical results.
// ------------------------------------------------------------------------------------------------------------------
template <int DIM, int N> struct LimBuf {
    // one workgroup per CU at the large orders (LDS): give it the waves.  cfg 4's shape (N = 8, 13 K patches), per launch:
    // project 3.76 / 2.90 / 2.46 ms and reconstruct 1.92 / 1.61 / 1.65 ms with 256 / 512 / 1024 threads (per-variable kernels: 5.41, 2.15)
    static constexpr int NT = (DIM == 3 && N >= 5) ? 1024 : 256;             // projection
    static constexpr int NT_R = (DIM == 3 && N >= 5) ? 512 : 256;            // reconstruction
    static constexpr int Ns = 2 * N - 1;
    static constexpr int NN = DIM == 3 ? N * N * N : N * N;
    static constexpr int B1 = DIM == 3 ? (NN > Ns * Ns * N ? NN : Ns * Ns * N) : Ns * N;  // project: in | after two passes; reconstruct: after one
    static constexpr int B2 = DIM == 3 ? Ns * N * N : Ns * N;                               // project: after one pass; reconstruct: after two
    static size_t bytes(int nv) { return sizeof(double) * (size_t)(B1 + B2) * nv; }
};

// out(t, acc[NV]) for t over (outer, nr, inner);  acc[v] = sum_c M[r0 + r][c] * in((o*C + c)*inner + in_i, v)
template <int C, int NV, class In, class Out>
__device__ inline void lim_apply_v(const double* M, int ldm, int r0, int nr, int dim, const int* e, int axis, In in, Out out) {
    unsigned inner = 1, outer = 1;
    for (int a = axis + 1; a < dim; a++) inner *= e[a];
    for (int a = 0; a < axis; a++) outer *= e[a];
    const unsigned total = outer * nr * inner;
    for (unsigned t = threadIdx.x; t < total; t += blockDim.x) {
        const unsigned q = t / inner, in_i = t - q * inner;
        const unsigned o = q / (unsigned)nr, r = q - o * nr;
        const double* mr = M + (r0 + r) * ldm;
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) acc[v] = 0.0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const double m = mr[c];
            const unsigned src = (o * C + c) * inner + in_i;
#pragma unroll
            for (int v = 0; v < NV; v++) acc[v] += m * in(src, v);
        }
        out(t, acc);
    }
}

template <int DIM, int N, int NV>
__global__ void __launch_bounds__((LimBuf<DIM, N>::NT))
limiter_project_all_kernel(long nc0, long nc1, long nc2, const double* __restrict__ u, const long* __restrict__ cells,
                           double* __restrict__ patch, const double* __restrict__ P_global, LimGhosts gh) {
    extern __shared__ __attribute__((aligned(16))) double lim_sm[];
    const double* P = P_global;
    using LB = LimBuf<DIM, N>;
    constexpr int Ns = LB::Ns, S = Ns + 2, NN = LB::NN;
    constexpr int SS = DIM == 3 ? S * S * S : S * S, LAYER = DIM == 3 ? Ns * Ns : Ns;
    double* X = lim_sm;
    double* Y = lim_sm + (size_t)LB::B1 * NV;
    __shared__ double Psh[Ns * N];                                 // the operator: every FMA reads an entry (a few distinct ones per wave)
    const long cell = cells[blockIdx.x];
    if (cell < 0) return;                                          // empty slot of a capacity-sized cell list
    for (int t = threadIdx.x; t < Ns * N; t += blockDim.x) Psh[t] = P[t];
    P = Psh;                                                       // (visible after the barrier behind the load of u)
    double* pt = patch + (long)blockIdx.x * SS * NV;
    const long nc[3] = {nc0, nc1, nc2};
    long cc[3];
    { long b = cell; cc[2] = DIM == 3 ? b % nc2 : 0; if (DIM == 3) b /= nc2; cc[1] = b % nc1; cc[0] = b / nc1; }
    auto fromX = [&](unsigned i, int v) { return X[i * NV + v]; };
    auto fromY = [&](unsigned i, int v) { return Y[i * NV + v]; };
    auto toX = [&](unsigned t, const double (&a)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; v++) X[t * NV + v] = a[v];
    };
    auto toY = [&](unsigned t, const double (&a)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; v++) Y[t * NV + v] = a[v];
    };
    // ---- own projection: u (AoS, contiguous) -> X -> Y (-> X) -> patch interior
    for (int t = threadIdx.x; t < NN * NV; t += blockDim.x) X[t] = u[cell * (long)(NN * NV) + t];
    __syncthreads();
    {
        int e[3] = {N, N, N};
        lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e, 0, fromX, toY);
        e[0] = Ns;
        __syncthreads();
        auto to_patch = [&](unsigned t, const double (&a)[NV]) {
            unsigned r = t;
            long flat = 0, mul = 1;
#pragma unroll
            for (int ax = DIM - 1; ax >= 0; ax--) { flat += (long)(r % Ns + 1) * mul; mul *= S; r /= Ns; }
#pragma unroll
            for (int v = 0; v < NV; v++) pt[flat * NV + v] = a[v];
        };
        if constexpr (DIM == 3) {
            lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e, 1, fromY, toX);
            e[1] = Ns;
            __syncthreads();
            lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e, 2, fromX, to_patch);
        } else {
            lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e, 1, fromY, to_patch);
        }
    }
    __syncthreads();                                               // the interior is in global memory, visible to the workgroup
    // edge and corner halo entries (>= 2 halo coordinates; never read by the 7-point stencil): nearest interior value
    for (int t = threadIdx.x; t < SS; t += blockDim.x) {
        int I[3] = {0, 0, 0}, r = t, nh = 0;
        for (int a = DIM - 1; a >= 0; a--) { I[a] = r % S; r /= S; nh += (I[a] == 0 || I[a] == S - 1) ? 1 : 0; }
        if (nh < 2) continue;
        long flat = 0;
        for (int a = 0; a < DIM; a++) {
            const int ci = I[a] < 1 ? 1 : (I[a] > Ns ? Ns : I[a]);
            flat = flat * S + ci;
        }
#pragma unroll
        for (int v = 0; v < NV; v++) pt[(long)t * NV + v] = pt[flat * NV + v];
    }
    // ---- face halos: the adjacent subcell layer of the face neighbours' projections
    for (int a = 0; a < DIM; a++)
        for (int side = 0; side < 2; side++) {
            auto to_layer = [&](unsigned t, const double (&acc)[NV]) {       // t enumerates the transverse subcells (axes != a, lexicographic)
                unsigned r = t;
                int idx[3] = {0, 0, 0};
                for (int b = DIM - 1; b >= 0; b--) {
                    if (b == a) continue;
                    idx[b] = r % Ns + 1;
                    r /= Ns;
                }
                idx[a] = side ? S - 1 : 0;
                long flat = 0;
                for (int b = 0; b < DIM; b++) flat = flat * S + idx[b];
#pragma unroll
                for (int v = 0; v < NV; v++) pt[flat * NV + v] = acc[v];
            };
            const double* g = gh.layer[a * 2 + side];
            if (g && cc[a] == (side ? nc[a] - 1 : 0)) {             // block boundary of a sharded grid: the layer arrived by exchange
                long tc = 0;
                for (int b = 0; b < DIM; b++)
                    if (b != a) tc = tc * nc[b] + cc[b];
                const double* gl = g + tc * LAYER * NV;
                for (int t = threadIdx.x; t < LAYER; t += blockDim.x) {
                    double acc[NV];
#pragma unroll
                    for (int v = 0; v < NV; v++) acc[v] = gl[(long)t * NV + v];
                    to_layer(t, acc);
                }
                continue;
            }
            long nb[3] = {cc[0], cc[1], cc[2]};
            nb[a] = (nb[a] + (side ? 1 : nc[a] - 1)) % nc[a];
            const long ncell = (nb[0] * nc1 + nb[1]) * (DIM == 3 ? nc2 : 1) + nb[2];
            __syncthreads();                                       // X and Y are free (the previous product has written its result)
            for (int t = threadIdx.x; t < NN * NV; t += blockDim.x) X[t] = u[ncell * (long)(NN * NV) + t];
            __syncthreads();
            int e2[3] = {N, N, N};
            const int b0 = a, b1 = (a + 1) % DIM, b2 = (a + 2) % DIM;
            lim_apply_v<N, NV>(P, N, side ? 0 : Ns - 1, 1, DIM, e2, b0, fromX, toY);      // normal axis first: the adjacent layer only
            e2[b0] = 1;
            __syncthreads();
            if constexpr (DIM == 3) {
                lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e2, b1, fromY, toX);
                e2[b1] = Ns;
                __syncthreads();
                lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e2, b2, fromX, to_layer);
            } else {
                lim_apply_v<N, NV>(P, N, 0, Ns, DIM, e2, b1, fromY, to_layer);
            }
        }
}

template <int DIM, int N, int NV>
__global__ void __launch_bounds__((LimBuf<DIM, N>::NT_R))
limiter_reconstruct_all_kernel(const double* __restrict__ patch, const long* __restrict__ cells, double* __restrict__ u,
                               const double* __restrict__ R_global) {
    extern __shared__ __attribute__((aligned(16))) double lim_sm[];
    const double* R = R_global;
    using LB = LimBuf<DIM, N>;
    constexpr int Ns = LB::Ns, S = Ns + 2, NN = LB::NN;
    constexpr int SS = DIM == 3 ? S * S * S : S * S;
    double* X = lim_sm;                                            // N_s^2 N (3-D) | N_s N (2-D): after one pass
    double* Y = lim_sm + (size_t)LB::B1 * NV;                      // N_s N^2: after two
    __shared__ double Rsh[N * Ns];
    const long cell = cells[blockIdx.x];
    if (cell < 0) return;
    for (int t = threadIdx.x; t < N * Ns; t += blockDim.x) Rsh[t] = R[t];
    __syncthreads();
    R = Rsh;
    const double* pt = patch + (long)blockIdx.x * SS * NV;
    auto from_patch = [&](unsigned i, int v) {                     // logical N_s^DIM index -> interior of the padded patch
        unsigned r = i;
        long flat = 0, mul = 1;
#pragma unroll
        for (int a = DIM - 1; a >= 0; a--) { flat += (long)(r % Ns + 1) * mul; mul *= S; r /= Ns; }
        return pt[flat * NV + v];
    };
    auto fromX = [&](unsigned i, int v) { return X[i * NV + v]; };
    auto fromY = [&](unsigned i, int v) { return Y[i * NV + v]; };
    auto toX = [&](unsigned t, const double (&a)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; v++) X[t * NV + v] = a[v];
    };
    auto toY = [&](unsigned t, const double (&a)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; v++) Y[t * NV + v] = a[v];
    };
    auto to_u = [&](unsigned t, const double (&a)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; v++) u[cell * (long)(NN * NV) + (long)t * NV + v] = a[v];
    };
    int e[3] = {Ns, Ns, Ns};
    lim_apply_v<Ns, NV>(R, Ns, 0, N, DIM, e, 0, from_patch, toX);
    e[0] = N;
    __syncthreads();
    if constexpr (DIM == 3) {
        lim_apply_v<Ns, NV>(R, Ns, 0, N, DIM, e, 1, fromX, toY);
        e[1] = N;
        __syncthreads();
        lim_apply_v<Ns, NV>(R, Ns, 0, N, DIM, e, 2, fromY, to_u);
    } else {
        lim_apply_v<Ns, NV>(R, Ns, 0, N, DIM, e, 1, fromX, to_u);
    }
}

// launch of the all-variables kernels (NV = 5: the Euler-sized systems of the configs); false: not built / does not fit -> per-variable path
template <int DIM, int N, int NV, int NT, class K, class... Args>
static bool lim_launch_all(K kern, long n, hipStream_t s, Args... args) {
    const size_t bytes = LimBuf<DIM, N>::bytes(NV);
    if (bytes + 2048 > 160 * 1024) return false;                   // (+ the operator copy and alignment)
    static bool attr_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!attr_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        attr_dev[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(NT), bytes, s, args...);
    return true;
}

#define EXA_LIM_CASES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

// EXA_LIM_PER_VARIABLE=1 in the environment routes five-variable systems through the per-variable kernels too (the path every
// other variable count takes): lets one process compare the two (tests/test_limiter.py).  Read at every call: it is a test switch.
static bool lim_per_variable() {
    const char* e = getenv("EXA_LIM_PER_VARIABLE");
    return e && e[0] == '1';
}

int limiter_project(int dim, int N, int Ns, int nv, const long* nc, const double* u, const long* cells, long n, double* patch,
                    const double* Pdev, const LimGhosts* ghosts, hipStream_t s) {
    if (n <= 0) return 0;
    LimGhosts gh{};
    if (ghosts) gh = *ghosts;
#ifndef EXA_LIM_PER_VARIABLE
    if (nv == 5 && !lim_per_variable()) {
        bool done = false;
        switch (N) {
#define X(NN_)                                                                                                                         \
        case NN_:                                                                                                                      \
            done = dim == 2 ? lim_launch_all<2, NN_, 5, LimBuf<2, NN_>::NT>(limiter_project_all_kernel<2, NN_, 5>, n, s, nc[0], nc[1], 1L, u, cells, patch, Pdev, gh) \
                            : lim_launch_all<3, NN_, 5, LimBuf<3, NN_>::NT>(limiter_project_all_kernel<3, NN_, 5>, n, s, nc[0], nc[1], nc[2], u, cells, patch, Pdev, gh); \
            break;
            EXA_LIM_CASES(X)
#undef X
        default: break;
        }
        if (done) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { set_error("limiter_project launch: %s", hipGetErrorString(e)); return -2; }
            return 0;
        }
    }
#endif
    switch (N) {
#define X(NN_)                                                                                                                         \
    case NN_:                                                                                                                          \
        if (dim == 2) hipLaunchKernelGGL((limiter_project_kernel<2, NN_>), dim3((unsigned)n), dim3(256), 0, s, N, Ns, nv, nc[0], nc[1], 1L, u, cells, patch, Pdev, gh); \
        else hipLaunchKernelGGL((limiter_project_kernel<3, NN_>), dim3((unsigned)n), dim3(256), 0, s, N, Ns, nv, nc[0], nc[1], nc[2], u, cells, patch, Pdev, gh);      \
        break;
        EXA_LIM_CASES(X)
#undef X
    default: set_error("limiter: N = %d is not built", N); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("limiter_project launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

int limiter_face_layers(int dim, int N, int Ns, int nv, const long* nc, const double* u, int a, int side, const double* need,
                        double* out, const double* Pdev, hipStream_t s) {
    long nt = 1;
    for (int b = 0; b < dim; b++)
        if (b != a) nt *= nc[b];
    if (nt <= 0) return 0;
    switch (N) {
#define X(NN_)                                                                                                                         \
    case NN_:                                                                                                                          \
        if (dim == 2) hipLaunchKernelGGL((limiter_face_layers_kernel<2, NN_>), dim3((unsigned)nt), dim3(256), 0, s, N, Ns, nv, nc[0], nc[1], 1L, u, a, side, need, out, Pdev); \
        else hipLaunchKernelGGL((limiter_face_layers_kernel<3, NN_>), dim3((unsigned)nt), dim3(256), 0, s, N, Ns, nv, nc[0], nc[1], nc[2], u, a, side, need, out, Pdev);      \
        break;
        EXA_LIM_CASES(X)
#undef X
    default: set_error("limiter: N = %d is not built", N); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("limiter_face_layers launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

int limiter_reconstruct(int dim, int N, int Ns, int nv, const double* patch, const long* cells, long n, double* u,
                        const double* Rdev, hipStream_t s) {
    if (n <= 0) return 0;
#ifndef EXA_LIM_PER_VARIABLE
    if (nv == 5 && !lim_per_variable()) {
        bool done = false;
        switch (N) {
#define X(NN_)                                                                                                                         \
        case NN_:                                                                                                                      \
            done = dim == 2 ? lim_launch_all<2, NN_, 5, LimBuf<2, NN_>::NT_R>(limiter_reconstruct_all_kernel<2, NN_, 5>, n, s, patch, cells, u, Rdev)        \
                            : lim_launch_all<3, NN_, 5, LimBuf<3, NN_>::NT_R>(limiter_reconstruct_all_kernel<3, NN_, 5>, n, s, patch, cells, u, Rdev);       \
            break;
            EXA_LIM_CASES(X)
#undef X
        default: break;
        }
        if (done) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { set_error("limiter_reconstruct launch: %s", hipGetErrorString(e)); return -2; }
            return 0;
        }
    }
#endif
    switch (N) {
#define X(NN_)                                                                                                                         \
    case NN_:                                                                                                                          \
        if (dim == 2) hipLaunchKernelGGL((limiter_reconstruct_kernel<2, NN_>), dim3((unsigned)n), dim3(256), 0, s, N, Ns, nv, patch, cells, u, Rdev); \
        else hipLaunchKernelGGL((limiter_reconstruct_kernel<3, NN_>), dim3((unsigned)n), dim3(256), 0, s, N, Ns, nv, patch, cells, u, Rdev);      \
        break;
        EXA_LIM_CASES(X)
#undef X
    default: set_error("limiter: N = %d is not built", N); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("limiter_reconstruct launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace exa
