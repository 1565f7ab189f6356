// FV subcell limiter glue for the ADER-DG path (BASELINE configs[4]; no counterpart in the
// reference).  Two small kernels around the fused FV Rusanov patch kernel (fv_rusanov.hip, the reference's kernel
// shape with patch_size = 2p+1, halo_size = 1):
//   limiter_project_kernel      troubled cell -> FV patch: interior = (P x P x P) u, face halos = the adjacent
//                               subcell layer of the face neighbours' projections (periodic in the block), edge
//                               and corner halo entries = nearest interior value (never read by the 7-point stencil)
//   limiter_reconstruct_kernel  FV patch interior -> DG nodes: u = (R x R x R) v
// One workgroup per troubled cell; the tensor products run variable by variable through two LDS buffers.  The kernels are
// instantiated per order (N, N_s compile-time): with run-time extents every output element paid three integer divisions
// (~100 instructions) for its 8 FMAs -- 7.6 ms per step for the 13 K patches of cfg 4's per-GPU shape, 8 % of the step.
#include <cstdio>
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

#define EXA_LIM_CASES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

int limiter_project(int dim, int N, int Ns, int nv, const long* nc, const double* u, const long* cells, long n, double* patch,
                    const double* Pdev, const LimGhosts* ghosts, hipStream_t s) {
    if (n <= 0) return 0;
    LimGhosts gh{};
    if (ghosts) gh = *ghosts;
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
