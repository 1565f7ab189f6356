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

#ifndef EXA_REG_PRIO
#define EXA_REG_PRIO 1
#endif
#ifndef EXA_REG_CPW
#define EXA_REG_CPW 2                          // cells in flight per workgroup: 2 = one 512-thread workgroup per CU, halves one barrier apart (128^3 cells:
#endif                                         // 154.5 ms; 1 = two 256-thread workgroups: 158.9 ms)
#define EXA_REG_CPW_DEFAULT EXA_REG_CPW

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

// CPW = cells a workgroup has in flight: 1 -> 256 threads, two workgroups per CU, each on its own; 2 -> 512 threads, one workgroup per
// CU whose two halves run the SAME program one barrier apart, so that one half's derive phase (arithmetic) always runs beside the
// other half's fold / load / closing phase (LDS traffic) -- the anti-phase two independent workgroups only reach by chance.
template <int N, class PDE, int CPW = 1> struct StageAReg {
    using G = RegGeo<N>;
    static constexpr int NV = PDE::NV, NA = PDE::NAUX;
    static constexpr int NT = 256;                                    // threads per cell in flight: one wave per SIMD
    static constexpr int LG = 2;                                      // time levels per step
    static constexpr int LS = (N + LG - 1) / LG;                      // steps per Picard iteration
    static constexpr int VS = LG * G::SL;                             // slot stride
    static constexpr int QSZ = NV * VS;                               // q (and each of the three sum arrays)
    // PAIRED (-DEXA_REG_PAIRED=1): the values of a node and level lie as 16-byte pairs (+ one 8-byte value where their number is odd), so a
    // pencil task reads the 7 values of a node with 3 ds_read_b128 + 1 ds_read_b64 instead of 7 ds_read_b64 and writes its 5 sums with
    // 2 + 1 stores; likewise the owners: 68 LDS instructions per step instead of 116, the same bytes.  Measured SLOWER (21.5 against
    // 19.4 ms per 64^3 launch, profiles/r03_reg_kernel.txt): SQ_INSTS_LDS -40 %, but SQ_WAIT_INST_LDS +24 % -- a 13-cycle ds_write_b128
    // holds the wave at the LDS queue longer than two 6-cycle stores interleaved with its arithmetic.  Off by default.
#ifndef EXA_REG_PAIRED
#define EXA_REG_PAIRED 0
#endif
    static constexpr bool PAIRED = EXA_REG_PAIRED != 0;
    static constexpr int NVA = NV + NA;
    static constexpr int PS2 = 2 * VS;                                // stride of a pair array
    static constexpr int SOFF = NVA * VS;                             // S_d at SOFF + d * QSZ
    // ncp term sets: three more arrays G_d = (D q) / h_d behind the sums -- the derive phase contracts q itself along its pencils, the node OWNER
    // evaluates B_d(q) G_d with the state it holds in registers (one cell per workgroup: 128 KB)
    // r5 (EXA_REG_NCP_ALIAS, default): the gradients take the place of the SUMS instead -- a step becomes [gradients] barrier [owners: B_d(q) G_d into
    // registers] barrier [flux sums] barrier [fold]; two more barriers per step, but the cell image is the 76 KB of a term set without ncp again, and two
    // cells are in flight per CU as everywhere else (one cell per CU = one wave per SIMD was the 2 x of profiles/r04_xt_ncp_kernels.txt)
#ifndef EXA_REG_NCP_ALIAS
#define EXA_REG_NCP_ALIAS 1
#endif
    static constexpr bool NCPV = pde_has_ncp<PDE>::value;
    static constexpr bool GALIAS = NCPV && EXA_REG_NCP_ALIAS != 0;
    static constexpr int GOFF = GALIAS ? SOFF : SOFF + 3 * QSZ;
    static constexpr int PIC_D = SOFF + ((NCPV && !GALIAS) ? 6 : 3) * QSZ;
    static constexpr int FS = G::NN;                                  // closing phases: [array][var][node], arrays qbar | Fbar_x | Fbar_y | Fbar_z (| source)
    static constexpr int FIN_D = (pde_has_source<PDE>::value ? 5 : 4) * NV * FS;
    static constexpr int CELL_D = PIC_D > FIN_D ? PIC_D : FIN_D;      // doubles of LDS per cell in flight
    static constexpr size_t LDS_BYTES = sizeof(double) * ((size_t)CELL_D * CPW + 2 * 3 * N);   // + the one-kernel step's corrector weights
    static constexpr bool FITS = G::NN <= NT && ((NCPV && !GALIAS) ? 1 : 2) * sizeof(double) * (size_t)CELL_D + 2048 <= 160 * 1024;
    // behind the image of dg_stage_a_kernel (StageA<3, N, PDE, CPB>::IMAGE_BYTES): lane -> packed derive task of a two-level
    // step, and of iteration 0 (one level); packed = d | level slot << 2 | pencil << 3, -1 = idle
    static constexpr int TAB_INTS = 2 * NT;
};

// Workgroup barrier.  -DEXA_REG_LDS_BARRIER: for LDS data only -- waits for this wave's LDS operations, not for its global loads / stores
// (the kernel hands nothing through global memory between its lanes, and __syncthreads() makes every wave wait for the acknowledgement
// of its trace stores).
__device__ inline void lds_barrier() {
#ifndef EXA_REG_LDS_BARRIER          // (measured: no difference, 19.5 against 19.4 ms per 64^3 launch -- the plain barrier stays)
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

// FUSE (the step as ONE kernel, "stage B folded into stage A"): the cell first finishes the PREVIOUS step -- Rusanov flux on its six
// faces from the traces that step left (own + neighbours', ghosts where the block ends; face-wide maximum eigenvalue as
// dg_stage_b_kernel) and the surface corrector on u* -- and runs the predictor on the result.  u* is updated in place (only this cell's
// own lanes touch it); the new traces go to a SECOND trace array, the neighbours still read the old one.  The 20 trace values of a
// face node (216 face nodes = 216 lanes) are requested during the previous cell's closing phases and stay in flight across its barriers
// (which wait for LDS operations only in this variant).  Sequence of a run: A, [B o A] x (n - 1), B.

// Term sets with a non-conservative product run ONE cell per workgroup at one wave per SIMD (REG_CPW_OF = 1, second launch bound 1): their three
// gradient arrays G_d = (D q) / h_d sit in LDS behind the sums (StageAReg::GOFF; 128 KB per cell), the node owners evaluate B_d(q) G_d.
// (First form: B_d(q_i) (D q)_i evaluated along the pencil in the derive phase -- six states at once on top of the owner state, ~290 VGPRs,
// 25 - 30 ms per 32^3 launch of Euler-with-pressure-as-ncp against 26.3 of the plain kernel; this form: 12.2 ms, profiles/r04_xt_ncp_kernels.txt.)
template <class PDE> constexpr int REG_CPW_OF = (pde_has_ncp<PDE>::value && !EXA_REG_NCP_ALIAS) ? 1 : EXA_REG_CPW_DEFAULT;
template <int N, class PDE, int CPW, bool FUSE = false>
__global__ void __launch_bounds__(256 * CPW, ((pde_has_ncp<PDE>::value && !EXA_REG_NCP_ALIAS) ? 1 : 2))
dg_stage_a_reg_kernel(const double* u_in, double* u_out, double* __restrict__ trace,
                      long ncells, CellBox box, double dt, double idx0, double idx1, double idx2, int n_it,
                      const void* __restrict__ ops_raw, const int* __restrict__ tab, const void* __restrict__ step_raw, RegFuse fz, PlainGeo geo) {
    // Term sets whose terms depend on position / time (XT) or carry a non-conservative product (NCP) -- generated term sets only
    // (pde_codegen.SympyPDE; the hooks of `Unit test/correctness_test.cpp:16-41,145-155`): node coordinates and level times reach the flux and
    // the source, B_d(q) (D q)_i / h_d joins the derivative sums of the pencil's nodes, the time-averaged ncp term is one more pass of the
    // derive phase over the final iterate.  Everything under `if constexpr`: the built-in term sets compile to the same code as before.
    constexpr bool XT = pde_has_xt<PDE>::value, NCP = pde_has_ncp<PDE>::value;
    constexpr bool FXT = pde_flux_xt<PDE>::value;                     // the flux itself sees x, t (otherwise only the source / ncp do)
    static_assert(!(FUSE && (XT || NCP)), "the one-kernel step is built for term sets without coordinates / ncp");
    using G = RegGeo<N>;
    using SA = StageAReg<N, PDE, CPW>;
    constexpr int NV = SA::NV, NA = SA::NA, DIM = 3;
    constexpr int NN = G::NN, NF = G::NF, SL = G::SL, PX = G::PX, PY = G::PY;
    constexpr int NT = SA::NT, LS = SA::LS, VS = SA::VS, QSZ = SA::QSZ, SOFF = SA::SOFF, FS = SA::FS, NVA = SA::NVA, PS2 = SA::PS2;
    [[maybe_unused]] constexpr int GOFF = SA::GOFF;
    constexpr int H = N / 2;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);  // which of the workgroup's cells in flight (wave-uniform)
    const int tid = threadIdx.x & 255;
    __attribute__((address_space(3))) double* lds = (__attribute__((address_space(3))) double*)lds_all + half * SA::CELL_D;
    [[maybe_unused]] const int bt = (tid & 63) + 64 * half, grp = tid >> 6;   // (stamp builds: one column per wave of the first cell in flight)
    EXA_STAMP_INIT();
    // workgroup barrier for LDS data only: the lanes hand nothing to each other through global memory, and the requests for the next cell (u,
    // FUSE: traces) and the u* stores stay in flight across it (__syncthreads() makes every wave wait for all of its vector memory operations)
    auto bar = [&]() {
#ifndef EXA_REG_FULL_BARRIER
#define EXA_REG_FULL_BARRIER 0
#endif
        if constexpr (FUSE && !EXA_REG_FULL_BARRIER) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        } else {
            lds_barrier();
        }
    };
    const bool node_lane = tid < NN;
    const int o_n = node_lane ? tid : 0;
    const int pk2 = tab[tid], pk1 = tab[NT + tid];

    typedef double d2_t __attribute__((ext_vector_type(2)));
    // K values of one node and level (element index idx = level slot * SL + node) in the array group at `base`
    auto ld_group = [&](int base, int idx, auto& out) {
        constexpr int K = sizeof(out) / sizeof(double);
        if constexpr (SA::PAIRED) {
#pragma unroll
            for (int u = 0; u < K / 2; u++) {
                const d2_t t2 = *(const __attribute__((address_space(3))) d2_t*)(&lds[base + u * PS2 + 2 * idx]);
                out[2 * u] = t2.x;
                out[2 * u + 1] = t2.y;
            }
            if constexpr (K % 2 == 1) out[K - 1] = EXA_LD(base + (K / 2) * PS2 + idx);
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) out[k] = EXA_LD(base + k * VS + idx);
        }
    };
    auto st_group = [&](int base, int idx, const auto& in) {
        constexpr int K = sizeof(in) / sizeof(double);
        if constexpr (SA::PAIRED) {
#pragma unroll
            for (int u = 0; u < K / 2; u++) *(__attribute__((address_space(3))) d2_t*)(&lds[base + u * PS2 + 2 * idx]) = d2_t{in[2 * u], in[2 * u + 1]};
            if constexpr (K % 2 == 1) EXA_ST(base + (K / 2) * PS2 + idx, in[K - 1]);
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) EXA_ST(base + k * VS + idx, in[k]);
        }
    };

    // ---- derive: one pencil task per lane, run-time direction.  s_i = sum_j D[i][j] (1/dx_d) f_d(q_j), even-odd form.
    // Em = DgOps::DEO in SGPRs (requested before the barrier in front of the phase).  The LDS loads of node pair j + 1 are issued
    // before the arithmetic of pair j: with two waves per SIMD nothing else hides the LDS latency of a wave.
    constexpr int NE = H * N + H + 1;
    struct Task {                                                      // one pencil task, decoded: first node, node stride (elements), array group of its sums
        int off, ps, so;
        bool on;
        int d, ls;                                                     // (XT / NCP: direction and level slot)
        double xa, xb;                                                 // (XT: reference coordinates xi of the pencil's two fixed node indices, axes ascending)
    };
    auto decode = [&](int pk, Task& k, DirFlux<PDE>& fx) {
        const int d = pk & 3, ls = (pk >> 2) & 1, t = pk >> 3;
        const int a = t / N, b = t - a * N;
        k.on = pk >= 0;
        k.off = ls * SL + (d == 0 ? a * PY + b : (d == 1 ? a * PX + b : a * PX + b * PY));
        k.ps = d == 0 ? PX : (d == 1 ? PY : 1);
        k.so = SOFF + d * QSZ;                                         // array group of this direction's sums
        k.d = d;
        k.ls = ls;
        if constexpr (XT) {
            k.xa = xi_of<N>(geo, a);
            k.xb = xi_of<N>(geo, b);
        }
        fx = DirFlux<PDE>(d, d == 0 ? idx0 : (d == 1 ? idx1 : idx2));
    };
    // XT: position of node jn of a pencil in the cell whose low corner is xc; time of the task's level slot
    [[maybe_unused]] double xc[3] = {0.0, 0.0, 0.0};                  // low corner of the current cell (wave-uniform)
    [[maybe_unused]] auto pencil_x = [&](const Task& tk, int d, int jn, double (&x)[3]) {
        const double xj = geo.xi[jn];
        x[0] = xc[0] + (d == 0 ? xj : tk.xa) * geo.h[0];
        x[1] = xc[1] + (d == 1 ? xj : (d == 0 ? tk.xa : tk.xb)) * geo.h[1];
        x[2] = xc[2] + (d == 2 ? xj : tk.xb) * geo.h[2];
    };
    // owner lane: reference coordinates of its node (once per kernel), position in the current cell
    [[maybe_unused]] double xio[3] = {0.0, 0.0, 0.0};
    if constexpr (XT) {
        xio[0] = xi_of<N>(geo, o_n / (N * N));
        xio[1] = xi_of<N>(geo, (o_n / N) % N);
        xio[2] = xi_of<N>(geo, o_n % N);
    }
    [[maybe_unused]] auto owner_x = [&](double (&x)[3]) {
#pragma unroll
        for (int a = 0; a < 3; a++) x[a] = xc[a] + xio[a] * geo.h[a];
    };
    [[maybe_unused]] auto level_t = [&](int l) -> double { return geo.t + geo.xi[l] * dt; };     // (l: compile-time at every call)
    [[maybe_unused]] auto source_at = [&](const double* qv, int l, double* S) {
        if constexpr (pde_has_source<PDE>::value) {
            if constexpr (XT) {
                double x[3];
                owner_x(x);
                PDE::source_xt(qv, x, level_t(l), S);
            } else {
                PDE::source(qv, S);
            }
        }
    };
    // the task of the two-level steps stays decoded in registers (30 integer instructions per step otherwise, 12 % of the phase)
    Task tk2;
    DirFlux<PDE> fx2(0, 0.0);
    decode(pk2, tk2, fx2);
    bool active_now = true;
    // first half: every node of the pencil is requested at once; flux at the nodes; the even / odd combinations e_j = F_j + F_{N-1-j},
    // o_j = F_j - F_{N-1-j} stay in registers
    static_assert(N % 2 == 0, "register-resident stage A: even N (the middle node of an odd N is not coded)");
    // tA / tB: times of the two level slots of the step (XT).  MODE 0: the derivative sums; 1: only the non-conservative part (the closing pass)
    // dc: the task's direction as a compile-time value (term sets whose flux has no per-lane-normal form: the caller splits by direction once
    // per task, so that the term calls see a constant normal), or -1: run-time normal through the masks of DirFlux
    auto derive_a = [&](auto dc, const Task& tk, const DirFlux<PDE>& fx, double (&e)[H][NV], double (&o)[H][NV], [[maybe_unused]] double tA,
                        [[maybe_unused]] double tB) {
        constexpr int DC = decltype(dc)::value;
        if (tk.on && active_now) {
            const int off = tk.off, ps = tk.ps;
            double qa[N][NVA];                                         // q | flux scalars of the six nodes
#pragma unroll
            for (int j = 0; j < H; j++) {
                ld_group(0, off + j * ps, qa[j]);
                ld_group(0, off + (N - 1 - j) * ps, qa[N - 1 - j]);
            }
            [[maybe_unused]] const double tl = tk.ls ? tB : tA;
            [[maybe_unused]] const int dd = DC >= 0 ? DC : tk.d;
            [[maybe_unused]] const double scd = dd == 0 ? idx0 : (dd == 1 ? idx1 : idx2);
#pragma unroll
            for (int j = 0; j < H; j++) {
                double Fa[NV], Fb[NV];
                if constexpr (FXT) {
                    double xj[3];
                    pencil_x(tk, dd, j, xj);
                    dg_flux_xt<PDE>(qa[j], xj, tl, dd, Fa);
                    pencil_x(tk, dd, N - 1 - j, xj);
                    dg_flux_xt<PDE>(qa[N - 1 - j], xj, tl, dd, Fb);
#pragma unroll
                    for (int v = 0; v < NV; v++) { Fa[v] *= scd; Fb[v] *= scd; }
                } else if constexpr (DC >= 0) {
                    PDE::template flux_scaled<DC>(qa[j], qa[j] + NV, scd, Fa);
                    PDE::template flux_scaled<DC>(qa[N - 1 - j], qa[N - 1 - j] + NV, scd, Fb);
                } else {
                    fx(qa[j], qa[j] + NV, Fa);
                    fx(qa[N - 1 - j], qa[N - 1 - j] + NV, Fb);
                }
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    e[j][v] = Fa[v] + Fb[v];
                    o[j][v] = Fa[v] - Fb[v];
                }
            }
        }
    };
    // NCP: G_d = (D q) / h_d along the pencil -- the SAME even-odd contraction as for the flux, on the states themselves (no term call, no
    // coordinates: those belong to the node owner, who evaluates B_d(q) G_d in the fold with the state it holds in registers); a pass of its
    // own in front of the flux part, so that the registers of the two passes overlap
    [[maybe_unused]] auto derive_g = [&](const Task& tk, [[maybe_unused]] const double (&Em)[NE]) {
        if constexpr (NCP) {
            if (tk.on && active_now) {
                const int off = tk.off, ps = tk.ps, go = GOFF + tk.d * QSZ;
                const double scd = tk.d == 0 ? idx0 : (tk.d == 1 ? idx1 : idx2);
                double e[H][NV], o[H][NV];
#pragma unroll
                for (int j = 0; j < H; j++) {
                    double qa[NV], qb[NV];
                    ld_group(0, off + j * ps, qa);
                    ld_group(0, off + (N - 1 - j) * ps, qb);
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        e[j][v] = scd * (qa[v] + qb[v]);
                        o[j][v] = scd * (qa[v] - qb[v]);
                    }
                }
#pragma unroll
                for (int i = 0; i < H; i++) {
                    double M[NV], gI[NV], gM[NV];
#pragma unroll
                    for (int v = 0; v < NV; v++) M[v] = Em[H + i] * o[0][v];
#pragma unroll
                    for (int j = 1; j < H; j++)
#pragma unroll
                        for (int v = 0; v < NV; v++) EXA_FMA(M[v], Em[j * N + H + i], o[j][v]);
#pragma unroll
                    for (int v = 0; v < NV; v++) gI[v] = M[v];
#pragma unroll
                    for (int j = 0; j < H; j++)
#pragma unroll
                        for (int v = 0; v < NV; v++) EXA_FMA(gI[v], Em[j * N + i], e[j][v]);
#pragma unroll
                    for (int v = 0; v < NV; v++) gM[v] = fma(2.0, M[v], -gI[v]);
                    st_group(go, off + i * ps, gI);
                    st_group(go, off + (N - 1 - i) * ps, gM);
                }
            }
        }
    };
    // ... and the owner's part: sum_d B_d(q) G_d at its node for level slot ls (q: the state of that level, l its index)
    [[maybe_unused]] auto ncp_at_owner = [&](const double* qv, int ls, int l, double (&out)[NV]) {
#pragma unroll
        for (int v = 0; v < NV; v++) out[v] = 0.0;
        if constexpr (NCP) {
            double xo[3] = {0.0, 0.0, 0.0};
            if constexpr (XT) owner_x(xo);
            static_for<0, DIM>([&](auto dc) {
                constexpr int D = decltype(dc)::value;
                double g[NV], nd[NV];
                ld_group(GOFF + D * QSZ, o_n + ls * SL, g);
#pragma unroll
                for (int v = 0; v < NV; v++) nd[v] = 0.0;
                dg_ncp<PDE>(qv, g, xo, XT ? level_t(l) : 0.0, D, nd);
#pragma unroll
                for (int v = 0; v < NV; v++) out[v] += nd[v];
            });
        }
    };
    // second half: output-stationary by row pair (i, N-1-i), each pair stored as soon as it is complete -- its stores drain under the
    // arithmetic of the next pair (stored all at the end, the 30 stores of each of the four waves queue up in front of the barrier)
    auto derive_b = [&](const Task& tk, const double (&Em)[NE], const double (&e)[H][NV], const double (&o)[H][NV]) {
        if (tk.on && active_now) {
            const int off = tk.off, ps = tk.ps, so = tk.so;
#pragma unroll
            for (int i = 0; i < H; i++) {
                // s_i = M + P, s_{N-1-i} = M - P with M = sum_j Eo[j][i] o_j, P = sum_j Ee[j][i] e_j: the P chain starts from M (no separate add),
                // the mirror row is 2 M - s_i (one FMA): 7 instead of 8 instructions per row pair and variable
                double M[NV], sI[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) M[v] = Em[H + i] * o[0][v];
#pragma unroll
                for (int j = 1; j < H; j++)
#pragma unroll
                    for (int v = 0; v < NV; v++) EXA_FMA(M[v], Em[j * N + H + i], o[j][v]);
#pragma unroll
                for (int v = 0; v < NV; v++) sI[v] = M[v];
#pragma unroll
                for (int j = 0; j < H; j++)
#pragma unroll
                    for (int v = 0; v < NV; v++) EXA_FMA(sI[v], Em[j * N + i], e[j][v]);
                double sM[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) sM[v] = fma(2.0, M[v], -sI[v]);
                st_group(so, off + i * ps, sI);
                st_group(so, off + (N - 1 - i) * ps, sM);
            }
        }
    };
    // (-DEXA_REG_SPLIT, CPW == 2: a barrier between the halves, i.e. three segments of similar length per step, the second cell in flight one
    //  segment behind the first -- measured 9 % SLOWER than two segments: a barrier costs more than the better balance returns,
    //  profiles/r03_reg_kernel.txt)
    using M0 = std::integral_constant<int, 0>;                       // derive: the derivative sums (with their gradients G_d where those have arrays of their own)
    using M1 = std::integral_constant<int, 1>;                       // ... only the gradients
    using M2 = std::integral_constant<int, 2>;                       // ... only the flux sums (GALIAS: the gradients were a phase of their own)
    constexpr bool GALIAS = SA::GALIAS;
    constexpr bool SPLIT_DIR = FXT || !pde_has_dir<PDE>::value;
    auto derive = [&](const Task& tk, const DirFlux<PDE>& fx, const double (&Em)[NE], double tA, double tB, auto mode) {
        // the derive stream is the long one of a step: it gets the SIMD's issue slots ahead of the co-resident wave of the other cell in
        // flight (in its fold / load / closing segment) -- 8 % of the launch (profiles/r03_reg_kernel.txt)
        __builtin_amdgcn_s_setprio(EXA_REG_PRIO);
        double e[H][NV], o[H][NV];
        constexpr int MODE = decltype(mode)::value;
        static_assert(!(GALIAS && MODE == 0), "gradients and sums share their arrays: two phases (M1, owners, M2)");
        if constexpr (MODE != 2) derive_g(tk, Em);
        auto front = [&](auto dc) {                                    // the direction-dependent part: term calls
            if constexpr (MODE != 1) derive_a(dc, tk, fx, e, o, tA, tB);
        };
        if constexpr (SPLIT_DIR) {
            // no per-lane-normal form of the flux (DirFlux<PDE, true>): the lanes of a wave take the branch of their direction, the term
            // calls inside see a compile-time normal (a wave with pencils of two directions runs two branches)
            if (tk.d == 0) front(std::integral_constant<int, 0>{});
            else if (tk.d == 1) front(std::integral_constant<int, 1>{});
            else front(std::integral_constant<int, 2>{});
        } else {
            front(std::integral_constant<int, -1>{});
        }
        if constexpr (MODE != 1) {
#ifdef EXA_REG_SPLIT
            if constexpr (CPW == 2) bar();
#endif
            derive_b(tk, Em, e, o);                                    // the contraction is the same for every direction
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- persistent grid: box slot -> cell, advanced incrementally (wave-uniform integers: no 64-bit division per cell).  Slot of this
    // half in trip k: (blockIdx.x * CPW + half) + k * gridDim.x * CPW; every half makes the same number of trips (the barriers are the
    // workgroup's), a half whose slot is past the end of the box idles through its last trip.
    int cx, cy, cz, sx, sy, sz;
    long trips;
    {
        const long b0 = (long)blockIdx.x * CPW + half, g = (long)gridDim.x * CPW;
        cz = (int)(b0 % box.nb[2]);
        cy = (int)((b0 / box.nb[2]) % box.nb[1]);
        cx = (int)(b0 / (box.nb[2] * box.nb[1]));
        sz = (int)(g % box.nb[2]);
        sy = (int)((g / box.nb[2]) % box.nb[1]);
        sx = (int)(g / (box.nb[2] * box.nb[1]));
        const long first = (long)blockIdx.x * CPW;
        trips = first < box.nbox ? (box.nbox - first + g - 1) / g : 0;
    }
    const int nb0 = (int)box.nb[0], nb1 = (int)box.nb[1], nb2 = (int)box.nb[2];
    auto cell_of = [&](int x, int y, int z) { return ((box.lo[0] + x) * box.nc[1] + box.lo[1] + y) * box.nc[2] + box.lo[2] + z; };
    double un[NV];                                                     // u of the NEXT cell's node, requested during this cell's closing phases
#pragma unroll
    for (int v = 0; v < NV; v++) un[v] = (node_lane && cx < nb0) ? u_in[(cell_of(cx, cy, cz) * NN + o_n) * NV + v] : 1.0;
    // FUSE: lane (face ff, face node fy) of the Riemann solve; minus / plus side states and normal fluxes of that node
    constexpr int TS = 2 * NV * NF;
    [[maybe_unused]] double tq[FUSE ? 4 : 1][NV];
    // minus / plus side trace blocks of this lane's face for cell (x, y, z) of the box (minus = the left cell's R trace, plus = the right cell's
    // L trace).  Everything derived from the lane id is recomputed here, behind an opaque copy: hoisted out of the cell loop these lane
    // constants live through the Picard iterations and spill.
    [[maybe_unused]] auto face_blocks = [&](int x, int y, int z, const double*& pm, const double*& pp, int& fyl) -> bool {
        const int lt = opaque_v(tid);
        const int f = lt / NF;
        fyl = lt - f * NF;
        const int d = f >> 1, side = f & 1;
        const int c0 = (int)box.lo[0] + x, c1 = (int)box.lo[1] + y, c2 = (int)box.lo[2] + z;
        const int n0 = (int)box.nc[0], n1 = (int)box.nc[1], n2 = (int)box.nc[2];
        const int cell = (c0 * n1 + c1) * n2 + c2;
        const int cst = d == 0 ? n1 * n2 : (d == 1 ? n2 : 1);
        const int ccd = d == 0 ? c0 : (d == 1 ? c1 : c2), ncd = d == 0 ? n0 : (d == 1 ? n1 : n2);
        const int tcell = d == 0 ? c1 * n2 + c2 : (d == 1 ? c0 * n2 + c2 : c0 * n1 + c1);
        const double* gh = f == 0 ? fz.ghost[0] : (f == 1 ? fz.ghost[1] : (f == 2 ? fz.ghost[2] : (f == 3 ? fz.ghost[3] : (f == 4 ? fz.ghost[4] : fz.ghost[5]))));
        const bool inner = side == 0 ? ccd > 0 : ccd < ncd - 1;
        const int nb = inner ? (side == 0 ? cell - cst : cell + cst) : (side == 0 ? cell + (ncd - 1) * cst : cell - (ncd - 1) * cst);
        const double* own = fz.trace_in + ((long)(d * 2 + side) * ncells + cell) * TS;
        const double* oth = (!inner && gh) ? gh + (long)tcell * TS : fz.trace_in + ((long)(d * 2 + 1 - side) * ncells + nb) * TS;
        pm = side == 0 ? oth : own;
        pp = side == 0 ? own : oth;
        return lt < 2 * DIM * NF && x < nb0;
    };
    // The 20 trace values of this lane's face node: requested in the closing phases of the cell before, needed by the cell's first instructions.
    // (The vector memory counter runs in order and the reload of a spilled register is such an operation: nothing may spill between here
    // and there, or the wave waits for these loads at the reload.)
    [[maybe_unused]] auto fetch_faces = [&](int x, int y, int z) {
        if constexpr (FUSE) {
            const double *pm, *pp;
            int fyl;
#ifdef EXA_FUSE_ABL_NOLOAD
            const bool on = face_blocks(x, y, z, pm, pp, fyl) && false;
#else
            const bool on = face_blocks(x, y, z, pm, pp, fyl);
#endif
#pragma unroll
            for (int v = 0; v < NV; v++) {
                tq[0][v] = on ? pm[v * NF + fyl] : 1.0;
                tq[1][v] = on ? pp[v * NF + fyl] : 1.0;
                tq[2][v] = on ? pm[(NV + v) * NF + fyl] : 0.0;
                tq[3][v] = on ? pp[(NV + v) * NF + fyl] : 0.0;
            }
        }
    };
    // corrector weights dt' / h_d * phi_{L,R}(xi_i) / w_i as a table in LDS behind the cells' images (read by the owners in the prologue: no
    // registers held through the Picard loop, no vector memory operation in front of the trace requests)
    constexpr int CWT = SA::CELL_D * CPW;
    if constexpr (FUSE) {
        if (threadIdx.x < 2 * DIM * N) {
            const DgOps<N>* og = reinterpret_cast<const DgOps<N>*>(ops_raw);
            const int t = threadIdx.x, D = t / (2 * N), sd = (t / N) & 1, i = t % N;
            const double idd = D == 0 ? idx0 : (D == 1 ? idx1 : idx2);
            ((__attribute__((address_space(3))) double*)lds_all)[CWT + t] = fz.dt_prev * idd * og->iw[i] * (sd == 0 ? og->phiL[i] : og->phiR[i]);
        }
    }
    fetch_faces(cx, cy, cz);
#ifndef EXA_REG_LOCKSTEP
    if constexpr (CPW == 2) {
        if (half == 1) bar();                                // the second half runs one barrier (= one phase) behind the first
    }
#endif

    // FUSE: u* of the cell just finished is stored during the next cell (see there); the plain predictor stores it at once and keeps
    // __syncthreads() (measured: deferring it there costs 1 %, 158.4 against 156.9 ms per 128^3 launch)
    constexpr bool DEFER = FUSE;
    [[maybe_unused]] double upend[NV];
    [[maybe_unused]] long pend_cell = -1;
#pragma unroll
    for (int v = 0; v < NV; v++) upend[v] = 0.0;
    for (long trip = 0; trip < trips; trip++) {
        const bool active = cx < nb0;                                  // (wave-uniform)
        active_now = active;
        const bool owner = node_lane && active;
        const long cell = active ? cell_of(cx, cy, cz) : 0;
        if constexpr (XT) {
            xc[0] = geo.x0[0] + (double)(box.lo[0] + cx) * geo.h[0];
            xc[1] = geo.x0[1] + (double)(box.lo[1] + cy) * geo.h[1];
            xc[2] = geo.x0[2] + (double)(box.lo[2] + cz) * geo.h[2];
        }
        cz += sz;
        if (cz >= nb2) { cz -= nb2; cy += 1; }
        cy += sy;
        if (cy >= nb1) { cy -= nb1; cx += 1; }
        cx += sx;

        double q[N][NV];
        double Em[NE];
        sload<NE>(ops_here<N>(ops_raw)->DEO, Em);

        // what the pencil tasks read of level slot ls at this owner's node: q and the cached flux scalars
        auto put_level = [&](int ls, const double (&qv)[NV]) {
            double qa[NVA];
#pragma unroll
            for (int v = 0; v < NV; v++) qa[v] = qv[v];
            PDE::aux_fast(qv, qa + NV);
            st_group(0, o_n + ls * SL, qa);
        };
        // ---- Picard iteration 0: the iterate is constant in time -- one level, row sums of T
#ifdef EXA_FUSE_ABL_NOPROLOGUE
        if constexpr (FUSE) {
            if (tq[0][0] == 1.2345e-300) un[0] += tq[1][1] + tq[2][2] + tq[3][3];
        }
        if constexpr (false) {
#else
        if constexpr (FUSE) {
#endif
            // ---- the previous step's Riemann solve and corrector for this cell (the sums' region of LDS is free until the first derive)
            constexpr int LAM = SOFF, FSO = SOFF + NT;
            const int lt = opaque_v(tid);                               // (lane constants recomputed: see face_blocks)
            const int ff = lt / NF, fy = lt - ff * NF;
            const bool face_lane = lt < 2 * DIM * NF;
            const int d = ff >> 1;
            const double lam = face_lane ? fmax(PDE::maxeig_fast(tq[0], d), PDE::maxeig_fast(tq[1], d)) : 0.0;
            EXA_ST(LAM + tid, lam);
            bar();
            if (face_lane) {
                // face-wide maximum: every lane of the face reads its NF values (broadcast reads, all in flight at once: registers are free here)
                double lv[NF];
#pragma unroll
                for (int y = 0; y < NF; y++) lv[y] = EXA_LD(LAM + ff * NF + y);
#pragma unroll
                for (int st = 1; st < NF; st *= 2)
#pragma unroll
                    for (int y = 0; y + st < NF; y += 2 * st) lv[y] = fmax(lv[y], lv[y + st]);
                const double sm = lv[0];
#pragma unroll
                for (int v = 0; v < NV; v++) EXA_ST(FSO + (ff * NV + v) * NF + fy, 0.5 * (tq[2][v] + tq[3][v]) - 0.5 * sm * (tq[1][v] - tq[0][v]));
            }
            bar();
            if (owner) {
                static_for<0, DIM>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    // face node of this volume node: the node index with the digit of direction D removed (Geo::face_index)
                    const int on_ = opaque_v(o_n);
                    const int i = (on_ / G::pstride(D)) % N;
                    const int yy = D == 0 ? on_ % (N * N) : (D == 1 ? (on_ / (N * N)) * N + on_ % N : on_ / N);
                    const __attribute__((address_space(3))) double* cwt = (const __attribute__((address_space(3))) double*)lds_all + CWT + D * 2 * N;
                    const double cwL = cwt[i], cwR = cwt[N + i];
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        const double fL = EXA_LD(FSO + ((2 * D + 0) * NV + v) * NF + yy), fR = EXA_LD(FSO + ((2 * D + 1) * NV + v) * NF + yy);
                        un[v] -= cwR * fR - cwL * fL;
                    }
                });
                if (fz.u_plain) {
#pragma unroll
                    for (int v = 0; v < NV; v++) fz.u_plain[(cell * NN + o_n) * NV + v] = un[v];
                }
            }
        }
        double ukeep[NV];                                              // u of this cell's node (iteration starts, u*): 10 VGPRs against a global re-read in front of a fold
#pragma unroll
        for (int v = 0; v < NV; v++) ukeep[v] = un[v];
        // u* of the PREVIOUS cell goes out here, behind the first use of what was requested for this cell: the vector memory counter runs
        // in order, so stores issued between those requests and their first use would have to be acknowledged first
        if constexpr (DEFER) {
            if (node_lane && pend_cell >= 0) {
                double* uo = u_out + (pend_cell * NN + o_n) * NV;
#pragma unroll
                for (int v = 0; v < NV; v++) uo[v] = upend[v];
            }
        }
        // (XT: the terms depend on the level time, so a constant iterate does not make the levels equal -- every iteration is a full one, started
        // from q_l = u as the scheme defines it)
        if constexpr (XT) {
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = owner ? un[v] : 1.0;
        } else {
            if (owner) put_level(0, un);
            EXA_STAMP(0);
            bar();
            EXA_STAMP(1);
            [[maybe_unused]] double nd0[NV];                               // GALIAS: the owner's ncp term of iteration 0
            {
                Task tk1;
                DirFlux<PDE> fx1(0, 0.0);
                decode(opaque_v(pk1), tk1, fx1);
                if constexpr (GALIAS) {                                    // gradients | owners | flux sums
                    derive(tk1, fx1, Em, 0.0, 0.0, M1{});
                    bar();
                    if (owner) ncp_at_owner(un, 0, 0, nd0);
                    bar();
                    derive(tk1, fx1, Em, 0.0, 0.0, M2{});
                } else {
                    derive(tk1, fx1, Em, 0.0, 0.0, M0{});
                }
            }
            EXA_STAMP(2);
            bar();
            EXA_STAMP(3);
            if (owner) {
                double S[NV], Ts[N];
                sload<N>(step_here<N>(step_raw)->Tsdt, Ts);
                {
                    double Sx[NV], Sy[NV], Sz[NV];
                    ld_group(SOFF, o_n, Sx);
                    ld_group(SOFF + QSZ, o_n, Sy);
                    ld_group(SOFF + 2 * QSZ, o_n, Sz);
    #pragma unroll
                    for (int v = 0; v < NV; v++) S[v] = Sx[v] + (Sy[v] + Sz[v]);
                }
                if constexpr (pde_has_source<PDE>::value) {                // q_t + div F = S(q)
                    double Sq[NV];
                    PDE::source(un, Sq);
    #pragma unroll
                    for (int v = 0; v < NV; v++) S[v] -= Sq[v];
                }
                if constexpr (NCP) {                                       // + sum_d B_d(q) (D q) / h_d
                    double nd[NV];
                    if constexpr (GALIAS) {
    #pragma unroll
                        for (int v = 0; v < NV; v++) nd[v] = nd0[v];
                    } else {
                        ncp_at_owner(un, 0, 0, nd);
                    }
    #pragma unroll
                    for (int v = 0; v < NV; v++) S[v] += nd[v];
                }
    #pragma unroll
                for (int l = 0; l < N; l++)
    #pragma unroll
                    for (int v = 0; v < NV; v++) q[l][v] = fma(Ts[l], S[v], un[v]);
            } else {
    #pragma unroll
                for (int l = 0; l < N; l++)
    #pragma unroll
                    for (int v = 0; v < NV; v++) q[l][v] = 1.0;
            }
            EXA_STAMP(4);

        }
        // ---- Picard iterations 1 .. n_it - 1.  A step = [load] barrier [derive] barrier [fold]; the load of the NEXT step is issued inside the
        // fold, between its LDS loads and its arithmetic (the levels it writes are in registers since the previous iteration, and an owner
        // rewrites only its own node, which nobody reads between these two barriers): the 14 stores per lane drain under the fold's FMAs.
        auto load_levels = [&](auto lc, const double (&qq)[N][NV]) {
            constexpr int l0 = decltype(lc)::value;
            constexpr int NL = (l0 + 1 < N) ? 2 : 1;
#pragma unroll
            for (int ls = 0; ls < NL; ls++) put_level(ls, qq[l0 + ls]);
        };
        constexpr int IT0 = XT ? 0 : 1;                                // first full iteration
        if (n_it > IT0 && owner) load_levels(std::integral_constant<int, 0>{}, q);
        // r5 (EXA_REG_DEFER_FOLD: 0 never, 1 term sets with an ncp or with terms that see x, t (default), 2 every term set): the time contraction of an iteration DEFERRED to its end, as in
        // exa_dg_m8.hpp -- a fold only keeps S_x + S_y + S_z (+ source / ncp terms) of its two levels; the owner state in front of the derive phases is
        // the levels not yet in LDS + the sums so far (30 doubles) instead of iterate + accumulators (60).  That is what lets the ncp variant, whose owners
        // hold their products over the flux phase, run two cells per CU under the 256-VGPR cap (108 spilled registers with the immediate fold).
#ifndef EXA_REG_DEFER_FOLD
#define EXA_REG_DEFER_FOLD 1
#endif
        // Measured (r5, profiles/r05_xt_ncp_kernels.txt): ncp 11.8 -> 10.1 ms per 32^3 launch (with the two cells per CU it allows), x,t source 6.09 -> 5.91;
        // the built-in Euler set LOSES 15 % (175.6 against 152.3 ms per 128^3 launch: no spills to win back, and the contraction of a whole iteration in ONE
        // fold unbalances the two cells in flight, whose folds run beside each other's derive phases) -- hence not for term sets of the state alone.
        constexpr bool DEFER_FOLD = EXA_REG_DEFER_FOLD == 2 || (EXA_REG_DEFER_FOLD == 1 && (NCP || XT));
        for (int it = IT0; it < n_it; it++) {
            double acc[N][NV];
            [[maybe_unused]] double Sk[DEFER_FOLD ? N : 1][NV];
            static_for<0, LS>([&](auto sc_) {
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = st * 2;
                constexpr int NL = (l0 + 1 < N) ? 2 : 1;
                sload<NE>(ops_here<N>(ops_raw)->DEO, Em);
                EXA_STAMP(5);
                bar();
                EXA_STAMP(6);
                static_assert(NL == 2, "odd N: the last step of an iteration has one level -- mask the tasks of level slot 1");
                [[maybe_unused]] double ndk[NL][NV];                   // GALIAS: the owner's ncp terms of the step's levels
                if constexpr (GALIAS) {                                // gradients | owners: B_d(q) G_d | flux sums (into the arrays the gradients held)
                    derive(tk2, fx2, Em, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M1{});
                    bar();
                    if (owner) {
#pragma unroll
                        for (int ls = 0; ls < NL; ls++) ncp_at_owner(q[l0 + ls], ls, l0 + ls, ndk[ls]);
                    }
                    bar();
                    derive(tk2, fx2, Em, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M2{});
                } else {
                    derive(tk2, fx2, Em, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M0{});
                }
                // what the fold needs from memory, requested in front of the barrier: -dt T[l'][l0 + ls] (l' fastest) and, where the
                // iteration starts its accumulators, u
                [[maybe_unused]] double Tm[NL * N], uu[NV];
                if constexpr (!DEFER_FOLD) {
#pragma unroll
                    for (int k = 0; k < NL * N; k++) Tm[k] = step_here<N>(step_raw)->TdtT[l0 * N + k];
                    if constexpr (st == 0) {
#pragma unroll
                        for (int v = 0; v < NV; v++) uu[v] = ukeep[v];
                    }
                }
                EXA_STAMP(7);
                bar();
                EXA_STAMP(8);
                if (owner) {
                    if constexpr (!DEFER_FOLD) spin(Tm);
                    double Sx[NL][NV], Sy[NL][NV], Sz[NL][NV];         // every load first
#pragma unroll
                    for (int ls = 0; ls < NL; ls++) {
                        ld_group(SOFF, o_n + ls * SL, Sx[ls]);
                        ld_group(SOFF + QSZ, o_n + ls * SL, Sy[ls]);
                        ld_group(SOFF + 2 * QSZ, o_n + ls * SL, Sz[ls]);
                    }
                    if constexpr (st + 1 < LS) load_levels(std::integral_constant<int, l0 + 2>{}, q);     // the next step's levels
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ls = 0; ls < NL; ls++)
#pragma unroll
                        for (int v = 0; v < NV; v++) Sx[ls][v] += Sy[ls][v] + Sz[ls][v];
                    if constexpr (pde_has_source<PDE>::value) {        // q_t + div F = S(q): evaluated here, on the iterate in the owner's registers
                        // (kept from the load to the fold, the sources of two steps cost 40 VGPRs of a kernel at its cap)
#pragma unroll
                        for (int ls = 0; ls < NL; ls++) {
                            double Sq[NV];
                            source_at(q[l0 + ls], l0 + ls, Sq);
#pragma unroll
                            for (int v = 0; v < NV; v++) Sx[ls][v] -= Sq[v];
                        }
                    }
                    if constexpr (NCP) {                               // + sum_d B_d(q_l) G_d: the gradients the derive phase left, the state from registers
#pragma unroll
                        for (int ls = 0; ls < NL; ls++) {
                            double nd[NV];
                            if constexpr (GALIAS) {
#pragma unroll
                                for (int v = 0; v < NV; v++) nd[v] = ndk[ls][v];
                            } else {
                                ncp_at_owner(q[l0 + ls], ls, l0 + ls, nd);
                            }
#pragma unroll
                            for (int v = 0; v < NV; v++) Sx[ls][v] += nd[v];
                        }
                    }
                    if constexpr (DEFER_FOLD) {
#pragma unroll
                        for (int ls = 0; ls < NL; ls++)
#pragma unroll
                            for (int v = 0; v < NV; v++) Sk[l0 + ls][v] = Sx[ls][v];
                        if constexpr (st + 1 == LS) {                  // every level's sums are there: q := u - dt T S, two input levels per batch of scalar operands
                            static_for<0, LS>([&](auto s2_) {
                                constexpr int s2 = decltype(s2_)::value;
                                constexpr int m0 = s2 * 2;
                                double T2[NL * N];
                                sload<NL * N>(step_here<N>(step_raw)->TdtT + m0 * N, T2);
#pragma unroll
                                for (int ls = 0; ls < NL; ls++)
#pragma unroll
                                    for (int lp = 0; lp < N; lp++)
#pragma unroll
                                        for (int v = 0; v < NV; v++)
                                            acc[lp][v] = fma(T2[ls * N + lp], Sk[m0 + ls][v], (s2 == 0 && ls == 0) ? ukeep[v] : acc[lp][v]);
                            });
                        }
                    } else {
#pragma unroll
                    for (int ls = 0; ls < NL; ls++)
#pragma unroll
                        for (int lp = 0; lp < N; lp++)
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                if constexpr (st == 0) acc[lp][v] = fma(Tm[ls * N + lp], Sx[ls][v], ls == 0 ? uu[v] : acc[lp][v]);
                                else acc[lp][v] = fma(Tm[ls * N + lp], Sx[ls][v], acc[lp][v]);
                            }
                    }
                    // the last step of an iteration completes the new iterate: its first levels go to LDS right away
                    if constexpr (st + 1 == LS) {
                        if (it + 1 < n_it) load_levels(std::integral_constant<int, 0>{}, acc);
                    }
                }
                EXA_STAMP(9);
            });
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = owner ? acc[l][v] : 1.0;
        }

        // ---- NCP: the time-averaged non-conservative term  sum_l w_l sum_d B_d(q_l) (D q_l) / h_d  of the FINAL iterate enters u* point-wise: one more
        // pass of the derive phase over the levels, gradients only, the owners evaluate the products and weight them with w_l
        [[maybe_unused]] double pw[NV];
        if constexpr (NCP) {
#pragma unroll
            for (int v = 0; v < NV; v++) pw[v] = 0.0;
            if (owner) load_levels(std::integral_constant<int, 0>{}, q);
            static_for<0, LS>([&](auto sc_) {
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = st * 2;
                sload<NE>(ops_here<N>(ops_raw)->DEO, Em);
                bar();
                derive(tk2, fx2, Em, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M1{});
                bar();
                if (owner) {
                    double wl[2];
                    sload<2>(ops_here<N>(ops_raw)->w + l0, wl);
#pragma unroll
                    for (int ls = 0; ls < 2; ls++) {
                        double nd[NV];
                        ncp_at_owner(q[l0 + ls], ls, l0 + ls, nd);
#pragma unroll
                        for (int v = 0; v < NV; v++) pw[v] = fma(wl[ls], nd[v], pw[v]);
                    }
                    if constexpr (st + 1 < LS) load_levels(std::integral_constant<int, l0 + 2>{}, q);
                }
            });
        }
        // ---- time averages (A.3): qbar | Fbar_x | Fbar_y | Fbar_z (| time-averaged source), node-major images of stride FS
        bar();                                               // every fold has read its sums: the closing image reuses the LDS
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
                double a[nz(NA)], F[NV];
                PDE::aux_fast(q[l], a);
#pragma unroll
                for (int v = 0; v < NV; v++) qb[v] += wm[l] * q[l][v];
                static_for<0, DIM>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    if constexpr (FXT) {
                        double xo[3];
                        owner_x(xo);
                        dg_flux_xt<PDE>(q[l], xo, level_t(l), D, F);
                    } else {
                        PDE::template flux<D>(q[l], a, F);
                    }
#pragma unroll
                    for (int v = 0; v < NV; v++) Fb[D][v] += wm[l] * F[v];
                });
                if constexpr (pde_has_source<PDE>::value) {
                    double Sq[NV];
                    source_at(q[l], l, Sq);
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
        bar();

        // the next cell's u for this lane (the registers of the iterate are free now); used by iteration 0 of the next cell
#pragma unroll
        for (int v = 0; v < NV; v++) un[v] = (node_lane && cx < nb0) ? u_in[(cell_of(cx, cy, cz) * NN + o_n) * NV + v] : 1.0;
        fetch_faces(cx, cy, cz);                                        // FUSE: the next cell's traces (tried earlier / later / with a touch pass: profiles/r03_one_kernel_step.txt)

        // ---- volume integral (in place over Fbar_d) + face extrapolation: one round per direction (compile-time strides), tasks (v, t), t fastest
        {
            double KE[NE], iwm[N], pl[N], pr[N];
            sload<NE>(ops_here<N>(ops_raw)->KEO, KE);
            sload<N>(ops_here<N>(ops_raw)->iw, iwm);
            sload<N>(ops_here<N>(ops_raw)->phiL, pl);
            sload<N>(ops_here<N>(ops_raw)->phiR, pr);
            static_assert(NV * NF <= NT, "closing phases: one round per direction");
            const int v = tid / NF, t = tid - v * NF;
            static_for<0, DIM>([&](auto dc) {
                constexpr int D = decltype(dc)::value;
                if (tid < NV * NF) {
                    constexpr int ps = G::pstride(D);
                    const int pb = G::pbase(D, t);
                    double qb[N], Fb[N], vol[N];
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        qb[j] = EXA_LD(v * FS + pb + j * ps);
                        Fb[j] = EXA_LD(((1 + D) * NV + v) * FS + pb + j * ps);
                    }
                    eo_apply<N>(KE, Fb, vol);
                    const double sc = dt * (D == 0 ? idx0 : (D == 1 ? idx1 : idx2));
#pragma unroll
                    for (int i = 0; i < N; i++) EXA_ST(((1 + D) * NV + v) * FS + pb + i * ps, sc * iwm[i] * vol[i]);
                    double qL = 0.0, qR = 0.0, FL = 0.0, FR = 0.0;
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        qL += pl[j] * qb[j];
                        qR += pr[j] * qb[j];
                        FL += pl[j] * Fb[j];
                        FR += pr[j] * Fb[j];
                    }
                    if (active) {
                        double* tl = trace + (((long)D * 2 + 0) * ncells + cell) * (2 * NV * NF);
                        double* tr = trace + (((long)D * 2 + 1) * ncells + cell) * (2 * NV * NF);
                        tl[(0 * NV + v) * NF + t] = qL;
                        tl[(1 * NV + v) * NF + t] = FL;
                        tr[(0 * NV + v) * NF + t] = qR;
                        tr[(1 * NV + v) * NF + t] = FR;
                    }
                }
            });
        }
        EXA_STAMP(11);
        bar();

        // ---- u* = u + sum_d vol_d (+ dt * time-averaged source): each owner its node (u is in registers; 40 contiguous bytes per lane)
        if (owner) {
            double us[NV];
#pragma unroll
            for (int v2 = 0; v2 < NV; v2++) {
                us[v2] = ukeep[v2];
                if constexpr (NCP) us[v2] -= dt * pw[v2];
                if constexpr (pde_has_source<PDE>::value) us[v2] += dt * EXA_LD((4 * NV + v2) * FS + o_n);
#pragma unroll
                for (int d = 0; d < DIM; d++) us[v2] += EXA_LD(((1 + d) * NV + v2) * FS + o_n);
            }
            if constexpr (DEFER) {
#pragma unroll
                for (int v2 = 0; v2 < NV; v2++) upend[v2] = us[v2];
            } else {
                double* uo = u_out + (cell * NN + o_n) * NV;
#pragma unroll
                for (int v2 = 0; v2 < NV; v2++) uo[v2] = us[v2];
            }
        }
        if constexpr (DEFER) pend_cell = active ? cell : -1;
        bar();                                               // LDS is reused by the next cell
    }
    if constexpr (DEFER) {
        if (node_lane && pend_cell >= 0) {
            double* uo = u_out + (pend_cell * NN + o_n) * NV;
#pragma unroll
            for (int v = 0; v < NV; v++) uo[v] = upend[v];
        }
    }
#ifndef EXA_REG_LOCKSTEP
    if constexpr (CPW == 2) {
        if (half == 0) bar();                                // (the barrier the second half is behind)
    }
#endif
    EXA_STAMP_FLUSH();
}

}  // namespace exa
