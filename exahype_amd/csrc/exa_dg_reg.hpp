// Stage A with the space-time iterate in REGISTERS ("register-resident", 3-D cells with N^3 <= 256 nodes; built for N = 6,
// the order of BASELINE configs[2]/[3]).  Same scheme and same results (to rounding) as dg_stage_a_kernel; no counterpart
// in the reference.
//
// Why a second kernel for the same order: dg_stage_a_kernel keeps the whole space-time image of a cell in LDS (155 KB), so a CU
// holds ONE workgroup whose arithmetic-bound phase (derivative sums) and LDS-store-bound phases (sum store, time update) are
// separated by workgroup barriers and add up (profiles/r02_stage_a_stamps.txt).  Here a node-owner lane keeps the iterate
// q_l(node) of all N time levels and the output-stationary accumulators of the time contraction in its registers, and the
// Picard iteration is streamed over the time levels two at a time (as exa_dg_stream.hpp does for N = 7, 8, but with nothing
// outside the chip: no slab).  LDS holds only two levels of q | flux scalars | S_x | S_y | S_z = 76 KB, a workgroup is 256
// threads at <= 256 VGPRs -- TWO workgroups share a CU and each one's arithmetic runs under the other's LDS traffic.
//
//   step (levels l0, l0 + 1):
//     load    owners put q_l and the cached flux scalars (Euler: 1/rho, p -- once per node and level, not per direction) into LDS
//     derive  216 pencil tasks (3 directions x 2 levels x 36 pencils), ONE per lane: the lanes of a wave hold pencils of
//             different directions (run-time stride and normal), flux at the 6 nodes, even-odd form of D (half the FMAs),
//             sums stored to S_d -- each direction has its own array, nothing is held over a barrier
//     fold    owners read S_x + S_y + S_z at their node and add -dt T[l'][l] S_l into the accumulators of all l'
//             (started from u: after the last step the accumulators ARE the new iterate)
//   two barriers per step; fold -> next load needs none (an owner rewrites only its own node, which no other lane reads
//   before the next barrier).
//
// LDS image: [slot][level slot][node], node stride 1, level stride SL = 217 (odd), slot stride 2 SL; slots = NV of q, NAUX
// scalars, 3 x NV sums.  Which lane takes which pencil comes from host-built tables behind the operator image
// (dg_inst.hip fill_reg_tables; model and search: scripts/reg_tables.py): six of the eight 32-lane groups hold pencils of ONE
// direction with pairwise distinct bank residues (the x pencils of a level are 36 consecutive doubles; the z pencils start
// on even doubles, so a group takes 16 of each level -- the odd level stride interleaves them), the remaining pencils share
// the last groups with a few 2-way conflicts.
#pragma once
#include "exa_dg_kernels.hpp"

namespace exa {

// ---- flux with a per-LANE normal ------------------------------------------------------------------------------------
// A PDE struct may provide `struct Dir`, `dir_init(Dir&, d, scale)` and `flux_scaled_dir(q, aux, dir, F)` (exa_pde.hpp: Euler
// selects the normal momentum and adds the pressure through per-lane masks, straight-line code).  Without them the three
// compile-time directions are dispatched by a switch (divergent inside a wave; generated term sets).
template <class P, class = void> struct pde_has_dir : std::false_type {};
template <class P> struct pde_has_dir<P, std::void_t<typename P::Dir>> : std::true_type {};

template <class PDE, bool HAS = pde_has_dir<PDE>::value> struct DirFlux;
template <class PDE> struct DirFlux<PDE, true> {
    typename PDE::Dir c;
    __device__ inline DirFlux(int d, double sc) { PDE::dir_init(c, d, sc); }
    __device__ inline void operator()(const double* q, const double* a, double* F) const { PDE::flux_scaled_dir(q, a, c, F); }
};
template <class PDE> struct DirFlux<PDE, false> {
    int d;
    double sc;
    __device__ inline DirFlux(int d_, double sc_) : d(d_), sc(sc_) {}
    __device__ inline void operator()(const double* q, const double* a, double* F) const {
        if (d == 0) PDE::template flux_scaled<0>(q, a, sc, F);
        else if (d == 1) PDE::template flux_scaled<1>(q, a, sc, F);
        else PDE::template flux_scaled<2>(q, a, sc, F);
    }
};

template <int N> struct RegGeo {
    static constexpr int NN = N * N * N, NF = N * N;
    static constexpr int PY = N, PX = N * N;                          // node stride 1, unpadded rows and planes
    static constexpr int SL = NN + (NN % 2 == 0 ? 1 : 0);             // level stride: odd
    __host__ __device__ static constexpr int pstride(int d) { return d == 0 ? PX : (d == 1 ? PY : 1); }
    __host__ __device__ static inline int pbase(int d, int t) {
        const int a = t / N, b = t - a * N;
        return d == 0 ? a * PY + b : (d == 1 ? a * PX + b : a * PX + b * PY);
    }
};

template <int N, class PDE> struct StageAReg {
    using G = RegGeo<N>;
    static constexpr int NV = PDE::NV, NA = PDE::NAUX;
    static constexpr int NT = 256;                                    // one wave per SIMD; two workgroups per CU
    static constexpr int LG = 2;                                      // time levels per step
    static constexpr int LS = (N + LG - 1) / LG;                      // steps per Picard iteration
    static constexpr int VS = LG * G::SL;                             // slot stride
    static constexpr int QSZ = NV * VS;                               // q (and each of the three sum arrays)
    static constexpr int AOFF = QSZ;                                  // flux scalars
    static constexpr int SOFF = (NV + NA) * VS;                       // S_d at SOFF + d * QSZ
    static constexpr int PIC_D = SOFF + 3 * QSZ;
    static constexpr int FS = G::NN;                                  // closing phases: [array][var][node], arrays qbar | Fbar_x | Fbar_y | Fbar_z (| source)
    static constexpr int FIN_D = (pde_has_source<PDE>::value ? 5 : 4) * NV * FS;
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)(PIC_D > FIN_D ? PIC_D : FIN_D);
    static constexpr bool FITS = G::NN <= NT && 2 * LDS_BYTES + 2048 <= 160 * 1024;
    // behind the image of dg_stage_a_kernel (StageA<3, N, PDE, CPB>::IMAGE_BYTES): lane -> packed derive task of a two-level
    // step, and of iteration 0 (one level); packed = d | level slot << 2 | pencil << 3, -1 = idle
    static constexpr int TAB_INTS = 2 * NT;
};

template <int N, class PDE>
__global__ void __launch_bounds__(256, 2)
dg_stage_a_reg_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, double* __restrict__ trace,
                      long ncells, CellBox box, double dt, double idx0, double idx1, double idx2, int n_it,
                      const void* __restrict__ ops_raw, const int* __restrict__ tab, const void* __restrict__ step_raw) {
    using G = RegGeo<N>;
    using SA = StageAReg<N, PDE>;
    constexpr int NV = SA::NV, NA = SA::NA, DIM = 3;
    constexpr int NN = G::NN, NF = G::NF, SL = G::SL, PX = G::PX, PY = G::PY;
    constexpr int NT = SA::NT, LS = SA::LS, VS = SA::VS, QSZ = SA::QSZ, AOFF = SA::AOFF, SOFF = SA::SOFF, FS = SA::FS;
    constexpr int H = N / 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];

    const int tid = threadIdx.x;
    [[maybe_unused]] const int bt = tid, grp = 0;                      // (stamp builds)
    EXA_STAMP_INIT();
    const bool owner = tid < NN;
    const int o_n = owner ? tid : 0;
    const int pk2 = tab[tid], pk1 = tab[NT + tid];

    // ---- derive: one pencil task per lane, run-time direction.  s_i = sum_j D[i][j] (1/dx_d) f_d(q_j), even-odd form.
    auto derive = [&](int pk_in, int nl) {
        const int pk = opaque_v(pk_in);                                // (decoded here, every time: nothing of it lives across the phases)
        const int ls = (pk >> 2) & 1;
        if (pk >= 0 && ls < nl) {
            const int d = pk & 3, t = pk >> 3;
            const int a = t / N, b = t - a * N;
            const int off = ls * SL + (d == 0 ? a * PY + b : (d == 1 ? a * PX + b : a * PX + b * PY));
            const int ps = d == 0 ? PX : (d == 1 ? PY : 1);
            const int so = off + SOFF + d * QSZ;                       // where the sums of this pencil go
            const DirFlux<PDE> fx(d, d == 0 ? idx0 : (d == 1 ? idx1 : idx2));
            const EXA_AS4 double* Em = ops_here<N>(ops_raw)->DEO;
            double P[H > 0 ? H : 1][NV], M[H > 0 ? H : 1][NV], mid[NV], s[N][NV];
#pragma unroll
            for (int i = 0; i < H; i++)
#pragma unroll
                for (int v = 0; v < NV; v++) P[i][v] = M[i][v] = 0.0;
#pragma unroll
            for (int v = 0; v < NV; v++) mid[v] = 0.0;
#pragma unroll
            for (int j = 0; j < H; j++) {
                const int ja = off + j * ps, jb = off + (N - 1 - j) * ps;
                double qa[NV], aa[NA], Fa[NV], qb[NV], ab[NA], Fb[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    qa[v] = EXA_LD(ja + v * VS);
                    qb[v] = EXA_LD(jb + v * VS);
                }
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    aa[k] = EXA_LD(ja + AOFF + k * VS);
                    ab[k] = EXA_LD(jb + AOFF + k * VS);
                }
                fx(qa, aa, Fa);
                fx(qb, ab, Fb);
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const double e = Fa[v] + Fb[v], o = Fa[v] - Fb[v];
                    Fa[v] = e;
                    Fb[v] = o;
                }
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const double ea = Em[j * N + i], eb = Em[j * N + H + i];
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        P[i][v] += ea * Fa[v];
                        M[i][v] += eb * Fb[v];
                    }
                }
                if constexpr (N % 2 == 1) {
                    const double em = Em[j * N + 2 * H];
#pragma unroll
                    for (int v = 0; v < NV; v++) mid[v] += em * Fb[v];
                }
            }
            if constexpr (N % 2 == 1) {                                // middle node
                const int jc = off + H * ps;
                double qa[NV], aa[NA], Fa[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) qa[v] = EXA_LD(jc + v * VS);
#pragma unroll
                for (int k = 0; k < NA; k++) aa[k] = EXA_LD(jc + AOFF + k * VS);
                fx(qa, aa, Fa);
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const double ec = Em[H * N + i];
#pragma unroll
                    for (int v = 0; v < NV; v++) P[i][v] += ec * Fa[v];
                }
#pragma unroll
                for (int v = 0; v < NV; v++) s[H][v] = mid[v];
            }
#pragma unroll
            for (int i = 0; i < H; i++)
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    s[i][v] = M[i][v] + P[i][v];
                    s[N - 1 - i][v] = M[i][v] - P[i][v];
                }
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int v = 0; v < NV; v++) EXA_ST(so + i * ps + v * VS, s[i][v]);
        }
    };

    // ---- persistent grid: box slot -> cell, advanced incrementally (wave-uniform integers: no 64-bit division per cell)
    int cx, cy, cz, sx, sy, sz;
    {
        const long b0 = blockIdx.x, g = gridDim.x;
        cz = (int)(b0 % box.nb[2]);
        cy = (int)((b0 / box.nb[2]) % box.nb[1]);
        cx = (int)(b0 / (box.nb[2] * box.nb[1]));
        sz = (int)(g % box.nb[2]);
        sy = (int)((g / box.nb[2]) % box.nb[1]);
        sx = (int)(g / (box.nb[2] * box.nb[1]));
    }
    const int nb0 = (int)box.nb[0], nb1 = (int)box.nb[1], nb2 = (int)box.nb[2];

    for (; cx < nb0;) {
        const long cell = ((box.lo[0] + cx) * box.nc[1] + box.lo[1] + cy) * box.nc[2] + box.lo[2] + cz;
        cz += sz;
        if (cz >= nb2) { cz -= nb2; cy += 1; }
        cy += sy;
        if (cy >= nb1) { cy -= nb1; cx += 1; }
        cx += sx;

        double u[NV], q[N][NV];
#pragma unroll
        for (int v = 0; v < NV; v++) u[v] = owner ? u_in[(cell * NN + o_n) * NV + v] : 1.0;

        // ---- Picard iteration 0: the iterate is constant in time -- one level, row sums of T
        if (owner) {
            double a[NA];
            PDE::aux_fast(u, a);
#pragma unroll
            for (int v = 0; v < NV; v++) EXA_ST(o_n + v * VS, u[v]);
#pragma unroll
            for (int k = 0; k < NA; k++) EXA_ST(o_n + AOFF + k * VS, a[k]);
        }
        EXA_STAMP(0);
        __syncthreads();
        EXA_STAMP(1);
        derive(pk1, 1);
        EXA_STAMP(2);
        __syncthreads();
        EXA_STAMP(3);
        if (owner) {
            double S[NV], Ts[N];
            sload<N>(step_here<N>(step_raw)->Tsdt, Ts);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const double sx_ = EXA_LD(o_n + SOFF + v * VS), sy_ = EXA_LD(o_n + SOFF + QSZ + v * VS), sz_ = EXA_LD(o_n + SOFF + 2 * QSZ + v * VS);
                S[v] = sx_ + sy_ + sz_;
            }
            if constexpr (pde_has_source<PDE>::value) {                // q_t + div F = S(q)
                double Sq[NV];
                PDE::source(u, Sq);
#pragma unroll
                for (int v = 0; v < NV; v++) S[v] -= Sq[v];
            }
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = fma(Ts[l], S[v], u[v]);
        } else {
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = 1.0;
        }
        EXA_STAMP(4);

        // ---- Picard iterations 1 .. n_it - 1
        for (int it = 1; it < n_it; it++) {
            double acc[N][NV];
            static_for<0, LS>([&](auto sc_) {
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = st * 2;
                constexpr int NL = (l0 + 1 < N) ? 2 : 1;
                [[maybe_unused]] double Sq[NL][NV];
                if (owner) {
#pragma unroll
                    for (int ls = 0; ls < NL; ls++) {
                        double a[NA];
                        PDE::aux_fast(q[l0 + ls], a);
#pragma unroll
                        for (int v = 0; v < NV; v++) EXA_ST(o_n + ls * SL + v * VS, q[l0 + ls][v]);
#pragma unroll
                        for (int k = 0; k < NA; k++) EXA_ST(o_n + ls * SL + AOFF + k * VS, a[k]);
                        if constexpr (pde_has_source<PDE>::value) PDE::source(q[l0 + ls], Sq[ls]);
                    }
                }
                EXA_STAMP(5);
                __syncthreads();
                EXA_STAMP(6);
                derive(pk2, NL);
                EXA_STAMP(7);
                __syncthreads();
                EXA_STAMP(8);
                if (owner) {
                    double Tm[NL * N];                                 // -dt T[l'][l0 + ls], l' fastest
                    sload<NL * N>(step_here<N>(step_raw)->TdtT + l0 * N, Tm);
                    double S[NL][NV];
#pragma unroll
                    for (int ls = 0; ls < NL; ls++)
#pragma unroll
                        for (int v = 0; v < NV; v++) {
                            const int p = o_n + ls * SL + SOFF + v * VS;
                            const double sx_ = EXA_LD(p), sy_ = EXA_LD(p + QSZ), sz_ = EXA_LD(p + 2 * QSZ);
                            S[ls][v] = sx_ + sy_ + sz_;
                            if constexpr (pde_has_source<PDE>::value) S[ls][v] -= Sq[ls][v];
                        }
#pragma unroll
                    for (int ls = 0; ls < NL; ls++)
#pragma unroll
                        for (int lp = 0; lp < N; lp++)
#pragma unroll
                            for (int v = 0; v < NV; v++)
                                acc[lp][v] = fma(Tm[ls * N + lp], S[ls][v], (st == 0 && ls == 0) ? u[v] : acc[lp][v]);
                }
                EXA_STAMP(9);
            });
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = owner ? acc[l][v] : 1.0;
        }

        // ---- time averages (A.3): qbar | Fbar_x | Fbar_y | Fbar_z (| time-averaged source), node-major images of stride FS
        __syncthreads();                                               // every fold has read its sums: the closing image reuses the LDS
        if (owner) {
            double wm[N];
            sload<N>(ops_here<N>(ops_raw)->w, wm);
            double qb[NV], Fb[DIM][NV];
            [[maybe_unused]] double Sbar[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) qb[v] = 0.0;
#pragma unroll
            for (int d = 0; d < DIM; d++)
#pragma unroll
                for (int v = 0; v < NV; v++) Fb[d][v] = 0.0;
            if constexpr (pde_has_source<PDE>::value) {
#pragma unroll
                for (int v = 0; v < NV; v++) Sbar[v] = 0.0;
            }
#pragma unroll
            for (int l = 0; l < N; l++) {
                double a[NA], F[NV];
                PDE::aux_fast(q[l], a);
#pragma unroll
                for (int v = 0; v < NV; v++) qb[v] += wm[l] * q[l][v];
                static_for<0, DIM>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    PDE::template flux<D>(q[l], a, F);
#pragma unroll
                    for (int v = 0; v < NV; v++) Fb[D][v] += wm[l] * F[v];
                });
                if constexpr (pde_has_source<PDE>::value) {
                    double Sq[NV];
                    PDE::source(q[l], Sq);
#pragma unroll
                    for (int v = 0; v < NV; v++) Sbar[v] += wm[l] * Sq[v];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                EXA_ST(v * FS + o_n, qb[v]);
#pragma unroll
                for (int d = 0; d < DIM; d++) EXA_ST(((1 + d) * NV + v) * FS + o_n, Fb[d][v]);
                if constexpr (pde_has_source<PDE>::value) EXA_ST((4 * NV + v) * FS + o_n, Sbar[v]);
            }
        }
        EXA_STAMP(10);
        __syncthreads();

        // ---- volume integral (in place over Fbar_d) + face extrapolation: pencil tasks (d, v, t), t fastest
        {
            constexpr int NE = H * N + H + 1;
            for (int task = tid; task < DIM * NV * NF; task += NT) {
                const int d = task / (NV * NF);
                const int r = task - d * (NV * NF);
                const int v = r / NF, t = r - v * NF;
                const int ps = G::pstride(d);
                const int pb = G::pbase(d, t);
                double KE[NE], iwm[N], pl[N], pr[N];
                sload<NE>(ops_here<N>(ops_raw)->KEO, KE);
                sload<N>(ops_here<N>(ops_raw)->iw, iwm);
                double qb[N], Fb[N], vol[N];
#pragma unroll
                for (int j = 0; j < N; j++) {
                    qb[j] = EXA_LD(v * FS + pb + j * ps);
                    Fb[j] = EXA_LD(((1 + d) * NV + v) * FS + pb + j * ps);
                }
                eo_apply<N>(KE, Fb, vol);
                const double sc = dt * (d == 0 ? idx0 : (d == 1 ? idx1 : idx2));
#pragma unroll
                for (int i = 0; i < N; i++) EXA_ST(((1 + d) * NV + v) * FS + pb + i * ps, sc * iwm[i] * vol[i]);
                sload<N>(ops_here<N>(ops_raw)->phiL, pl);
                sload<N>(ops_here<N>(ops_raw)->phiR, pr);
                double qL = 0.0, qR = 0.0, FL = 0.0, FR = 0.0;
#pragma unroll
                for (int j = 0; j < N; j++) {
                    qL += pl[j] * qb[j];
                    qR += pr[j] * qb[j];
                    FL += pl[j] * Fb[j];
                    FR += pr[j] * Fb[j];
                }
                double* tl = trace + (((long)d * 2 + 0) * ncells + cell) * (2 * NV * NF);
                double* tr = trace + (((long)d * 2 + 1) * ncells + cell) * (2 * NV * NF);
                tl[(0 * NV + v) * NF + t] = qL;
                tl[(1 * NV + v) * NF + t] = FL;
                tr[(0 * NV + v) * NF + t] = qR;
                tr[(1 * NV + v) * NF + t] = FR;
            }
        }
        EXA_STAMP(11);
        __syncthreads();

        // ---- u* = u + sum_d vol_d (+ dt * time-averaged source), AoS (coalesced)
        for (int e = tid; e < NN * NV; e += NT) {
            const int n = e / NV, v = e - n * NV;
            double us = u_in[cell * (NN * NV) + e];
            if constexpr (pde_has_source<PDE>::value) us += dt * EXA_LD((4 * NV + v) * FS + n);
#pragma unroll
            for (int d = 0; d < DIM; d++) us += EXA_LD(((1 + d) * NV + v) * FS + n);
            u_out[cell * (NN * NV) + e] = us;
        }
        __syncthreads();                                               // LDS is reused by the next cell
    }
    EXA_STAMP_FLUSH();
}

}  // namespace exa
