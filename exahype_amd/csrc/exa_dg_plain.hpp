// ADER-DG stage A for term sets whose terms depend on position / time (HAS_XT) or that carry a non-conservative product (HAS_NCP):
//   q_t + div F(q, x, t) + B(q, x, t) . grad q = S(q, x, t)
// -- the hooks the reference's harness declares for its kernel (`Unit test/correctness_test.cpp:16-41,145-155`: flux / ncp / source /
// eigenvalue with (Q, x, h, t, dt, ...)); no ADER-DG counterpart in the reference.
//
// A PLAIN kernel: one workgroup per cell, two space-time images (iterate; flux of the current direction, then the right-hand side) and u in
// LDS, every thread loops over space-time nodes: per direction the flux once per node, then the derivative sums.  It serves generated term sets
// (pde_codegen.SympyPDE) only; the built-in term sets never come here and their kernels carry none of this.  Not tuned -- correctness and
// the same data layouts (u*, traces) as the other stage-A kernels, so that stage B and the sharded step work unchanged.
//
// Scheme (restated in numpy for the tests: step_xt): node x = origin + (cell + xi_i) h, level time t_l = t + xi_l dt;
//   predictor   S_l = sum_a [ D F_a(q_l, x, t_l) + B_a(q_l, x, t_l) (D q_l) ] / h_a - S(q_l, x, t_l),   q_l' <- u - dt sum_l T[l'][l] S_l
//   averages    qbar, Fbar_a over the levels; the non-conservative term and the source enter u* point-wise:
//               u* = u + volume(Fbar) - dt sum_l w_l sum_a B_a(q_l)(D q_l)/h_a + dt sum_l w_l S(q_l)
//   traces      of qbar and Fbar_a as everywhere else.
#pragma once
#include "exa_dg_common.hpp"
#include "exa_pde.hpp"

namespace exa {

template <int DIM, int N, class PDE> struct StagePlain {
    static constexpr int NV = PDE::NV;
    static constexpr int NN = ipow(N, DIM), NF = ipow(N, DIM - 1);
    static constexpr int NT = 256;
    static constexpr int IMG = N * NN * NV;                           // one space-time image
    // second region: the right-hand side (N levels) during the Picard loop, then qbar | Fbar_a = 1 + DIM node arrays -- more than N of them
    // at N < DIM + 1 (3-D N = 2, 3; 2-D N = 2), so the region is sized for the larger of the two and u sits behind it
    static constexpr int BIMG = (N > 1 + DIM ? N : 1 + DIM) * NN * NV;
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)(IMG + BIMG + NN * NV);
    static constexpr bool FITS = LDS_BYTES <= 160 * 1024 && NN <= NT;
};

template <int DIM, int N, class PDE>
__global__ void __launch_bounds__(256)
dg_stage_a_plain_kernel(const double* u_in, double* u_out, double* __restrict__ trace, long ncells, CellBox box, double dt, double idx0,
                        double idx1, double idx2, int n_it, DgOps<N> ops, PlainGeo geo) {
    using SP = StagePlain<DIM, N, PDE>;
    constexpr int NV = SP::NV, NN = SP::NN, NF = SP::NF, NT = SP::NT, IMG = SP::IMG;
    extern __shared__ __attribute__((aligned(16))) double plain_lds[];
    double* A = plain_lds;              // iterate      [level][node][var]
    double* B = plain_lds + IMG;        // right-hand side, later qbar | Fbar_a  ([array][node][var])
    double* U = plain_lds + IMG + SP::BIMG;    // u     [node][var]
    const int tid = threadIdx.x;
    const long cell = box.cell(blockIdx.x);
    if (cell < 0) return;
    long cc[3];
    {
        long b = blockIdx.x;
        const long cz = b % box.nb[2];
        b /= box.nb[2];
        cc[0] = box.lo[0] + b / box.nb[1];
        cc[1] = box.lo[1] + b % box.nb[1];
        cc[2] = box.lo[2] + cz;
    }
    const double idx[3] = {idx0, idx1, idx2};
    auto stride = [](int a) { return a == DIM - 1 ? 1 : (a == DIM - 2 ? N : N * N); };
    auto digit = [&](int n, int a) { return (n / stride(a)) % N; };
    auto coords = [&](int n, double* x) {
#pragma unroll
        for (int a = 0; a < 3; a++) x[a] = a < DIM ? geo.x0[a] + ((double)cc[a] + geo.xi[digit(n, a)]) * geo.h[a] : 0.0;
    };

    for (int e = tid; e < NN * NV; e += NT) U[e] = u_in[cell * (NN * NV) + e];
    __syncthreads();
    for (int e = tid; e < N * NN * NV; e += NT) A[e] = U[e % (NN * NV)];
    __syncthreads();

    // B . grad q of one space-time node (grad = D q / h along each direction), summed over the directions
    auto ncp_of = [&](int l, int n, const double* x, double tl, double* out) {
        const double* ql = A + ((long)l * NN + n) * NV;
#pragma unroll
        for (int v = 0; v < NV; v++) out[v] = 0.0;
        if constexpr (pde_has_ncp<PDE>::value) {
            for (int a = 0; a < DIM; a++) {
                const int ia = digit(n, a);
                double grad[NV], o1[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) { grad[v] = 0.0; o1[v] = 0.0; }
                for (int j = 0; j < N; j++) {
                    const double* qj = A + ((long)l * NN + n + (j - ia) * stride(a)) * NV;
                    const double dij = ops.D[ia * N + j] * idx[a];
#pragma unroll
                    for (int v = 0; v < NV; v++) grad[v] += dij * qj[v];
                }
                fv_ncp<PDE>(ql, grad, x, tl, a, o1);
#pragma unroll
                for (int v = 0; v < NV; v++) out[v] += o1[v];
            }
        }
    };

    // Picard iterations.  Per direction: the flux of every space-time node into B (ONE evaluation per node and direction), then each thread
    // adds the derivative along that direction to the right-hand sides of its nodes (registers); the complete right-hand sides go back
    // into B for the time update.
    constexpr int KE = (N * NN + NT - 1) / NT;
    for (int it = 0; it < n_it; it++) {
        double rhs[KE][NV];
#pragma unroll
        for (int k = 0; k < KE; k++)
#pragma unroll
            for (int v = 0; v < NV; v++) rhs[k][v] = 0.0;
        for (int a = 0; a < DIM; a++) {
#pragma unroll
            for (int k = 0; k < KE; k++) {
                const int e = tid + k * NT;
                if (e < N * NN) {
                    const int l = e / NN, n = e - l * NN;
                    double x[3], F[NV];
                    coords(n, x);
#pragma unroll
                    for (int v = 0; v < NV; v++) F[v] = 0.0;
                    fv_flux<PDE>(A + (long)e * NV, x, geo.t + geo.xi[l] * dt, a, F);
#pragma unroll
                    for (int v = 0; v < NV; v++) B[(long)e * NV + v] = F[v];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KE; k++) {
                const int e = tid + k * NT;
                if (e < N * NN) {
                    const int l = e / NN, n = e - l * NN, ia = digit(n, a);
                    for (int j = 0; j < N; j++) {
                        const double dij = ops.D[ia * N + j] * idx[a];
                        const double* Fj = B + ((long)l * NN + n + (j - ia) * stride(a)) * NV;
#pragma unroll
                        for (int v = 0; v < NV; v++) rhs[k][v] += dij * Fj[v];
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < KE; k++) {
            const int e = tid + k * NT;
            if (e < N * NN) {
                const int l = e / NN, n = e - l * NN;
                const double tl = geo.t + geo.xi[l] * dt;
                double x[3];
                coords(n, x);
                if constexpr (pde_has_ncp<PDE>::value) {
                    double o1[NV];
                    ncp_of(l, n, x, tl, o1);
#pragma unroll
                    for (int v = 0; v < NV; v++) rhs[k][v] += o1[v];
                }
                if constexpr (pde_has_source<PDE>::value) {
                    double Sq[NV];
#pragma unroll
                    for (int v = 0; v < NV; v++) Sq[v] = 0.0;
                    fv_source<PDE>(A + (long)e * NV, x, tl, Sq);
#pragma unroll
                    for (int v = 0; v < NV; v++) rhs[k][v] -= Sq[v];
                }
#pragma unroll
                for (int v = 0; v < NV; v++) B[(long)e * NV + v] = rhs[k][v];
            }
        }
        __syncthreads();
        for (int e = tid; e < N * NN; e += NT) {
            const int lp = e / NN, n = e - lp * NN;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                double s = 0.0;
                for (int l = 0; l < N; l++) s += ops.T[lp * N + l] * B[((long)l * NN + n) * NV + v];
                A[(long)e * NV + v] = U[n * NV + v] - dt * s;
            }
        }
        __syncthreads();
    }

    // ---- time averages: every thread its node (NN <= NT); qbar | Fbar_a into B, the point-wise terms into `up`
    double up[NV], qb[NV], Fb[DIM][NV];
    const int n = tid;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        up[v] = 0.0;
        qb[v] = 0.0;
#pragma unroll
        for (int a = 0; a < DIM; a++) Fb[a][v] = 0.0;
    }
    if (n < NN) {
        double x[3], pw[NV];
        coords(n, x);
#pragma unroll
        for (int v = 0; v < NV; v++) pw[v] = 0.0;
        for (int l = 0; l < N; l++) {
            const double tl = geo.t + geo.xi[l] * dt;
            const double* ql = A + ((long)l * NN + n) * NV;
            for (int a = 0; a < DIM; a++) {
                double F[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) F[v] = 0.0;
                fv_flux<PDE>(ql, x, tl, a, F);
#pragma unroll
                for (int v = 0; v < NV; v++) Fb[a][v] += ops.w[l] * F[v];
            }
#pragma unroll
            for (int v = 0; v < NV; v++) qb[v] += ops.w[l] * ql[v];
            if constexpr (pde_has_ncp<PDE>::value) {
                double ncps[NV];
                ncp_of(l, n, x, tl, ncps);
#pragma unroll
                for (int v = 0; v < NV; v++) pw[v] -= ops.w[l] * ncps[v];
            }
            if constexpr (pde_has_source<PDE>::value) {
                double Sq[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) Sq[v] = 0.0;
                fv_source<PDE>(ql, x, tl, Sq);
#pragma unroll
                for (int v = 0; v < NV; v++) pw[v] += ops.w[l] * Sq[v];
            }
        }
#pragma unroll
        for (int v = 0; v < NV; v++) up[v] = U[n * NV + v] + dt * pw[v];
    }
    __syncthreads();                                                     // (B is free since the last update; nothing above writes LDS)
    if (n < NN) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            B[((long)0 * NN + n) * NV + v] = qb[v];
#pragma unroll
            for (int a = 0; a < DIM; a++) B[((long)(1 + a) * NN + n) * NV + v] = Fb[a][v];
        }
    }
    __syncthreads();

    // ---- volume integral: u* = up + sum_a dt / h_a / w_i sum_j Kxi[i][j] Fbar_a(node with index j along a)
    if (n < NN) {
        for (int a = 0; a < DIM; a++) {
            const int ia = digit(n, a);
            const double sc = dt * idx[a] * ops.iw[ia];
            for (int j = 0; j < N; j++) {
                const int m = n + (j - ia) * stride(a);
                const double k = sc * ops.Kxi[ia * N + j];
#pragma unroll
                for (int v = 0; v < NV; v++) up[v] += k * B[((long)(1 + a) * NN + m) * NV + v];
            }
        }
#pragma unroll
        for (int v = 0; v < NV; v++) u_out[(cell * NN + n) * NV + v] = up[v];
    }
    // ---- face extrapolation: tasks (direction, face node, variable)
    for (int e = tid; e < DIM * NF * NV; e += NT) {
        const int a = e / (NF * NV), r = e - a * (NF * NV), v = r / NF, y = r - v * NF;
        // first node of the pencil along a whose transverse index is y (lexicographic over the remaining axes)
        int base;
        if constexpr (DIM == 3) base = a == 0 ? y : (a == 1 ? (y / N) * N * N + y % N : y * N);
        else base = a == 0 ? y : y * N;
        double qL = 0.0, qR = 0.0, FL = 0.0, FR = 0.0;
        for (int j = 0; j < N; j++) {
            const int m = base + j * stride(a);
            const double q = B[((long)0 * NN + m) * NV + v], F = B[((long)(1 + a) * NN + m) * NV + v];
            qL += ops.phiL[j] * q;
            qR += ops.phiR[j] * q;
            FL += ops.phiL[j] * F;
            FR += ops.phiR[j] * F;
        }
        double* tl = trace + (((long)a * 2 + 0) * ncells + cell) * (2 * NV * NF);
        double* tr = trace + (((long)a * 2 + 1) * ncells + cell) * (2 * NV * NF);
        tl[(0 * NV + v) * NF + y] = qL;
        tl[(1 * NV + v) * NF + y] = FL;
        tr[(0 * NV + v) * NF + y] = qR;
        tr[(1 * NV + v) * NF + y] = FR;
    }
}

}  // namespace exa
