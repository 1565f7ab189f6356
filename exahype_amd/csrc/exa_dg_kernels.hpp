// ADER-DG cell kernels for gfx950 (MI355X).  No counterpart in the reference
// (SURVEY.md F2); the scheme is SURVEY.md Appendix A, the point-wise terms are
// the device twins of `Unit test/Functions.cpp` (exa_pde.hpp).
//
// Stage A  (dg_stage_a_kernel): per cell -- space-time predictor (Picard),
//          time averages, volume integral, face extrapolation.
// Stage B  (dg_stage_b_kernel): per cell -- Rusanov flux on its 2*DIM faces
//          (face-wide max eigenvalue) and the surface corrector.
//
// Mapping to the machine: CPB cells per workgroup, the whole space-time DoF block
// of a cell lives in LDS as SoA [var][time slab][padded node] (Geo<>), the 1-D
// operators arrive in SGPRs through the kernarg segment (DgOps<>), all
// contractions are sum-factorised pencil products kept in registers
// (N loads -> N*N fp64 FMAs per variable).  HBM traffic per cell is the
// compulsory one: u read once, u* written once, traces written once.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "exa_dg_common.hpp"
#include "exa_pde.hpp"

namespace exa {

template <int I, int E, class F> __device__ inline void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// ------------------------------------------------------------------------------------------
// Stage A
// ------------------------------------------------------------------------------------------
template <int DIM, int N, class PDE, int CPB> struct StageA {
    using G = Geo<DIM, N>;
    static constexpr int NV = PDE::NV;
    static constexpr int NA = PDE::NAUX;
    static constexpr int ASZ = NV * G::NTS * G::SL;         // one q-sized array
    static constexpr int AXO = 2 * ASZ;                      // aux offset inside a cell image
    static constexpr int CS = 2 * ASZ + NA * N * G::SL;      // doubles per cell image
    static constexpr size_t LDS_BYTES = (size_t)CPB * CS * sizeof(double);
};

template <int DIM, int N, class PDE, int CPB, int NT>
__global__ void __launch_bounds__(NT)
dg_stage_a_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, double* __restrict__ trace,
                  long ncells, CellBox box, double dt, double idx0, double idx1, double idx2, int n_it, DgOps<N> ops) {
    using G = Geo<DIM, N>;
    using SA = StageA<DIM, N, PDE, CPB>;
    constexpr int NV = PDE::NV, NA = PDE::NAUX;
    constexpr int NN = G::NN, NF = G::NF, SL = G::SL, NTS = G::NTS;
    constexpr int ASZ = SA::ASZ, AXO = SA::AXO, CS = SA::CS;
    constexpr int KMAX = (CPB * NN + NT - 1) / NT;           // node tasks per thread
    extern __shared__ __attribute__((aligned(16))) double lds[];

    const int tid = threadIdx.x;
    const long b0 = (long)blockIdx.x * CPB;       // first box slot of this workgroup
    const double idx[3] = {idx0, idx1, idx2};
    // box slot -> cell of the local block (-1: past the end of the box)
    auto cell_of = [&](int c) -> long { return box.cell(b0 + c); };

    // ---- load u (AoS, coalesced: consecutive lanes -> consecutive nodes), keep it in registers
    double ur[KMAX][NV];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int task = tid + k * NT;
        const int c = task / NN, n = task - c * NN;
        const long cell = task < CPB * NN ? cell_of(c) : -1;
#pragma unroll
        for (int v = 0; v < NV; v++) ur[k][v] = cell >= 0 ? u_in[(cell * NN + n) * NV + v] : 1.0;
        if (task < CPB * NN && n_it > 0) {
            const int off = c * CS + G::node_off(n);
            double a[NA];
            PDE::aux(ur[k], a);
#pragma unroll
            for (int l = 0; l < N; l++) {
#pragma unroll
                for (int v = 0; v < NV; v++) lds[off + (v * NTS + l) * SL] = ur[k][v];
#pragma unroll
                for (int i = 0; i < NA; i++) lds[off + AXO + (i * N + l) * SL] = a[i];
            }
        }
    }
    __syncthreads();

    // ---- Picard iterations (A.2):  q <- u - dt * T * sum_d (1/dx_d) D_d f_d(q)
    for (int it = 0; it < n_it; it++) {
        static_for<0, DIM>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            constexpr int ps = G::pstride(D);
            for (int task = tid; task < CPB * N * NF; task += NT) {
                const int c = task / (N * NF), r = task - c * (N * NF);
                const int l = r / NF, t = r - l * NF;
                const int off = c * CS + l * SL + G::pbase(D, t);
                double F[N][NV];
#pragma unroll
                for (int j = 0; j < N; j++) {
                    double q[NV], a[NA];
#pragma unroll
                    for (int v = 0; v < NV; v++) q[v] = lds[off + v * NTS * SL + j * ps];
#pragma unroll
                    for (int i = 0; i < NA; i++) a[i] = lds[off + AXO + i * N * SL + j * ps];
                    PDE::template flux<D>(q, a, F[j]);
                }
#pragma unroll
                for (int i = 0; i < N; i++) {
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        double s = 0.0;
#pragma unroll
                        for (int j = 0; j < N; j++) s += ops.D[i * N + j] * F[j][v];
                        s *= idx[D];
                        double* dst = &lds[off + ASZ + v * NTS * SL + i * ps];
                        if constexpr (D == 0) *dst = s;
                        else *dst += s;
                    }
                }
            }
            __syncthreads();
        });
        // time contraction per node, new cached scalars
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const int task = tid + k * NT;
            if (task < CPB * NN) {
                const int c = task / NN, n = task - c * NN;
                const int off = c * CS + G::node_off(n);
                double qn[N][NV];
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    double S[N];
#pragma unroll
                    for (int l = 0; l < N; l++) S[l] = lds[off + ASZ + (v * NTS + l) * SL];
#pragma unroll
                    for (int lp = 0; lp < N; lp++) {
                        double acc = 0.0;
#pragma unroll
                        for (int l = 0; l < N; l++) acc += ops.T[lp * N + l] * S[l];
                        qn[lp][v] = ur[k][v] - dt * acc;
                        lds[off + (v * NTS + lp) * SL] = qn[lp][v];
                    }
                }
#pragma unroll
                for (int lp = 0; lp < N; lp++) {
                    double a[NA];
                    PDE::aux(qn[lp], a);
#pragma unroll
                    for (int i = 0; i < NA; i++) lds[off + AXO + (i * N + lp) * SL] = a[i];
                }
            }
        }
        __syncthreads();
    }

    // ---- time averages (A.3) per node: qbar -> B slab 0, Fbar_d -> B slab 1+d, u -> A slab 0
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int task = tid + k * NT;
        if (task < CPB * NN) {
            const int c = task / NN, n = task - c * NN;
            const int off = c * CS + G::node_off(n);
            double qb[NV], Fb[DIM][NV];
#pragma unroll
            for (int v = 0; v < NV; v++) qb[v] = 0.0;
#pragma unroll
            for (int d = 0; d < DIM; d++)
#pragma unroll
                for (int v = 0; v < NV; v++) Fb[d][v] = 0.0;
            if (n_it > 0) {
#pragma unroll
                for (int l = 0; l < N; l++) {
                    double q[NV], a[NA], F[NV];
#pragma unroll
                    for (int v = 0; v < NV; v++) q[v] = lds[off + (v * NTS + l) * SL];
#pragma unroll
                    for (int i = 0; i < NA; i++) a[i] = lds[off + AXO + (i * N + l) * SL];
#pragma unroll
                    for (int v = 0; v < NV; v++) qb[v] += ops.w[l] * q[v];
                    static_for<0, DIM>([&](auto dc) {
                        constexpr int D = decltype(dc)::value;
                        PDE::template flux<D>(q, a, F);
#pragma unroll
                        for (int v = 0; v < NV; v++) Fb[D][v] += ops.w[l] * F[v];
                    });
                }
            } else {
                double a[NA];
                PDE::aux(ur[k], a);
#pragma unroll
                for (int v = 0; v < NV; v++) qb[v] = ur[k][v];
                static_for<0, DIM>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    PDE::template flux<D>(ur[k], a, Fb[D]);
                });
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                lds[off + ASZ + (v * NTS + 0) * SL] = qb[v];
#pragma unroll
                for (int d = 0; d < DIM; d++) lds[off + ASZ + (v * NTS + 1 + d) * SL] = Fb[d][v];
                lds[off + (v * NTS + 0) * SL] = ur[k][v];
            }
        }
    }
    __syncthreads();

    // ---- volume integral + face extrapolation: pencil tasks (c, d, v, t), t fastest
    for (int task = tid; task < CPB * DIM * NV * NF; task += NT) {
        const int c = task / (DIM * NV * NF);
        int r = task - c * (DIM * NV * NF);
        const int d = r / (NV * NF);
        r -= d * (NV * NF);
        const int v = r / NF, t = r - v * NF;
        const int ps = G::pstride(d);
        const int off = c * CS + G::pbase(d, t) + v * NTS * SL;
        double qb[N], Fb[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            qb[j] = lds[off + ASZ + j * ps];
            Fb[j] = lds[off + ASZ + (1 + d) * SL + j * ps];
        }
        const double sc = dt * idx[d];
#pragma unroll
        for (int i = 0; i < N; i++) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < N; j++) s += ops.Kxi[i * N + j] * Fb[j];
            lds[off + (1 + d) * SL + i * ps] = sc * ops.iw[i] * s;
        }
        double qL = 0.0, qR = 0.0, FL = 0.0, FR = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            qL += ops.phiL[j] * qb[j];
            qR += ops.phiR[j] * qb[j];
            FL += ops.phiL[j] * Fb[j];
            FR += ops.phiR[j] * Fb[j];
        }
        const long cell = cell_of(c);
        if (cell >= 0) {
            double* tl = trace + (((long)d * 2 + 0) * ncells + cell) * (2 * NV * NF);
            double* tr = trace + (((long)d * 2 + 1) * ncells + cell) * (2 * NV * NF);
            tl[(0 * NV + v) * NF + t] = qL;
            tl[(1 * NV + v) * NF + t] = FL;
            tr[(0 * NV + v) * NF + t] = qR;
            tr[(1 * NV + v) * NF + t] = FR;
        }
    }
    __syncthreads();

    // ---- u* = u + sum_d vol_d, written AoS (coalesced)
    for (int task = tid; task < CPB * NN * NV; task += NT) {
        const int c = task / (NN * NV), e = task - c * (NN * NV);
        const int n = e / NV, v = e - n * NV;
        const int off = c * CS + G::node_off(n) + v * NTS * SL;
        double us = lds[off];
#pragma unroll
        for (int d = 0; d < DIM; d++) us += lds[off + (1 + d) * SL];
        const long cell = cell_of(c);
        if (cell >= 0) u_out[cell * (NN * NV) + e] = us;
    }
}

// ------------------------------------------------------------------------------------------
// Stage B
// ------------------------------------------------------------------------------------------
struct StageBArgs {
    long nc[3];            // local block
    long lo[3], nb[3];     // box origin and extent
    const double* ghost[6];
};

template <int DIM, int N, class PDE, int CPB, int NT>
__global__ void __launch_bounds__(NT)
dg_stage_b_kernel(double* __restrict__ u, const double* __restrict__ trace, StageBArgs A, long ncells, long nbox,
                  double dt, double idx0, double idx1, double idx2, DgOps<N> ops) {
    using G = Geo<DIM, N>;
    constexpr int NV = PDE::NV;
    constexpr int NN = G::NN, NF = G::NF;
    constexpr int NFACE = 2 * DIM;
    constexpr int TS = 2 * NV * NF;                     // doubles per (cell, d, side) trace
    constexpr int KB = (CPB * NFACE * NF + NT - 1) / NT;
    __shared__ double lam[CPB * NFACE * NF];
    __shared__ double fs[CPB * NFACE * NF * NV];
    __shared__ double cL[DIM][N], cR[DIM][N];

    const int tid = threadIdx.x;
    const long b0 = (long)blockIdx.x * CPB;
    const double idx[3] = {idx0, idx1, idx2};
    // (compile-time index into the kernarg-resident operator block: stays in SGPRs)
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (tid == i) {
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                cL[d][i] = dt * idx[d] * ops.iw[i] * ops.phiL[i];
                cR[d][i] = dt * idx[d] * ops.iw[i] * ops.phiR[i];
            }
        }
    }

    // cell of box slot c
    auto cell_of = [&](int c, long* cc) -> long {
        long b = b0 + c;
        if (b >= nbox) b = nbox - 1;
        long cz = 0, cy, cx;
        if constexpr (DIM == 3) {
            cz = b % A.nb[2];
            b /= A.nb[2];
        }
        cy = b % A.nb[1];
        cx = b / A.nb[1];
        cc[0] = A.lo[0] + cx;
        cc[1] = A.lo[1] + cy;
        cc[2] = (DIM == 3) ? A.lo[2] + cz : 0;
        return (cc[0] * A.nc[1] + cc[1]) * A.nc[2] + cc[2];
    };

    double qm[KB][NV], qp[KB][NV], Fm[KB][NV], Fp[KB][NV];
#pragma unroll
    for (int k = 0; k < KB; k++) {
        const int task = tid + k * NT;
        if (task < CPB * NFACE * NF) {
            const int c = task / (NFACE * NF);
            int r = task - c * (NFACE * NF);
            const int f = r / NF, y = r - f * NF;
            const int d = f >> 1, face = f & 1;
            long cc[3];
            const long cell = cell_of(c, cc);
            long cst[3];
            cst[2] = 1;
            cst[1] = A.nc[2];
            cst[0] = A.nc[1] * A.nc[2];
            // transverse cell index on the block face (lexicographic over the other axes)
            long tcell;
            if constexpr (DIM == 3) tcell = d == 0 ? cc[1] * A.nc[2] + cc[2] : (d == 1 ? cc[0] * A.nc[2] + cc[2] : cc[0] * A.nc[1] + cc[1]);
            else tcell = d == 0 ? cc[1] : cc[0];
            const double *pm, *pp;
            if (face == 0) {   // minus = left neighbour's R trace, plus = own L trace
                pp = trace + (((long)d * 2 + 0) * ncells + cell) * TS;
                if (cc[d] > 0) pm = trace + (((long)d * 2 + 1) * ncells + cell - cst[d]) * TS;
                else if (A.ghost[d * 2 + 0]) pm = A.ghost[d * 2 + 0] + tcell * TS;
                else pm = trace + (((long)d * 2 + 1) * ncells + cell + (A.nc[d] - 1) * cst[d]) * TS;
            } else {           // minus = own R trace, plus = right neighbour's L trace
                pm = trace + (((long)d * 2 + 1) * ncells + cell) * TS;
                if (cc[d] < A.nc[d] - 1) pp = trace + (((long)d * 2 + 0) * ncells + cell + cst[d]) * TS;
                else if (A.ghost[d * 2 + 1]) pp = A.ghost[d * 2 + 1] + tcell * TS;
                else pp = trace + (((long)d * 2 + 0) * ncells + cell - (A.nc[d] - 1) * cst[d]) * TS;
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                qm[k][v] = pm[v * NF + y];
                qp[k][v] = pp[v * NF + y];
                Fm[k][v] = pm[(NV + v) * NF + y];
                Fp[k][v] = pp[(NV + v) * NF + y];
            }
            lam[task] = fmax(PDE::maxeig(qm[k], d), PDE::maxeig(qp[k], d));
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KB; k++) {
        const int task = tid + k * NT;
        if (task < CPB * NFACE * NF) {
            const int face0 = (task / NF) * NF;
            double s = 0.0;
            for (int y = 0; y < NF; y++) s = fmax(s, lam[face0 + y]);
#pragma unroll
            for (int v = 0; v < NV; v++)
                fs[task * NV + v] = 0.5 * (Fm[k][v] + Fp[k][v]) - 0.5 * s * (qp[k][v] - qm[k][v]);
        }
    }
    __syncthreads();
    for (int task = tid; task < CPB * NN * NV; task += NT) {
        const int c = task / (NN * NV), e = task - c * (NN * NV);
        if (b0 + c >= nbox) continue;
        const int n = e / NV, v = e - n * NV;
        long cc[3];
        const long cell = cell_of(c, cc);
        double un = u[cell * (NN * NV) + e];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            const int i = G::coord(n, d), y = G::face_index(n, d);
            const double fR = fs[((c * NFACE + d * 2 + 1) * NF + y) * NV + v];
            const double fL = fs[((c * NFACE + d * 2 + 0) * NF + y) * NV + v];
            un -= cR[d][i] * fR - cL[d][i] * fL;
        }
        u[cell * (NN * NV) + e] = un;
    }
}

// max eigenvalue over all nodes / directions (CFL)
template <int DIM, class PDE>
__global__ void dg_maxeig_kernel(const double* __restrict__ u, long nnodes, double* __restrict__ out) {
    constexpr int NV = PDE::NV;
    double m = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nnodes; i += (long)gridDim.x * blockDim.x) {
        double q[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) q[v] = u[i * NV + v];
#pragma unroll
        for (int d = 0; d < DIM; d++) m = fmax(m, PDE::maxeig(q, d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __shared__ double wm[16];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = fmax(m, wm[w]);
        // non-negative doubles order like their bit patterns -> integer atomic max
        atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(m));
    }
}

}  // namespace exa
