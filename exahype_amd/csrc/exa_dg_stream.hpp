// Stage A for the 3-D cells whose space-time image does not fit the LDS (N = 7, 8: p = 6, 7;
// the image alone is 160 KiB at N = 8).  Same scheme and same results (to rounding) as
// dg_stage_a_kernel; no counterpart in the reference.
//
// "Level-streamed" Picard loop.  The derivative sums S_l = sum_d (1/dx_d) D_d f_d(q_l) of a time level l
// need only q_l, and the time update q_l' = u - dt sum_l T[l'][l] S_l is linear in them.  So an iteration
// walks over the time levels two at a time:
//   load    q_l of the two levels (node owners), flux scalars once per node -> LDS
//   derive  the three directions at once, one pencil per lane, even-odd form of D:  y -> A, z -> B, x holds and
//           stores over Q after the barrier
//   fold    node owners add T[l'][l] * (S_x + S_y + S_z) into register accumulators for their output levels --
//           the output-stationary form of the time contraction
// and ends with q_l' = u - dt * acc, written back to the slab (the last iteration keeps it in registers for
// the time averages).  A step takes the levels g and LH + g, so that the lane that loads a level is the lane
// that produced it: slab traffic is lane-private (no fence), fetched one step ahead of its use, and the first
// level of each half never leaves the registers.  Per iteration and cell that is 3/4 of q written and read
// (120 KiB each way at N = 8) instead of every partial sum (the first, all-in-global-memory version); LDS holds two
// levels, not eight.
//
// LDS strides: odd row stride (N = 8: 9) and plane stride = 8 mod 32 doubles keep the y and z pencils
// conflict-free for 32-lane ds_read_b64 groups (scripts/lds_stride_search.py model); an owner-slot -> node permutation
// (kernel prologue) does the same for the node-linear phases; the x pencils keep a 2-way conflict.
#pragma once
#include "exa_dg_kernels.hpp"

// Diagnostic ablations (never in the product build): -DEXA_ABL_SKIP_D / _LOAD / _FOLD / _SLAB drop a phase to time the rest.
#ifdef EXA_ABL_SKIP_D
#define EXA_ABL_COND_D && n_it < 0
#else
#define EXA_ABL_COND_D
#endif
#ifdef EXA_ABL_SKIP_LOAD
#define EXA_ABL_COND_LOAD && n_it < 0
#else
#define EXA_ABL_COND_LOAD
#endif
#ifdef EXA_ABL_SKIP_FOLD
#define EXA_ABL_COND_FOLD && n_it < 0
#else
#define EXA_ABL_COND_FOLD
#endif
#ifdef EXA_ABL_SKIP_SLAB
#define EXA_ABL_COND_SLAB && n_it < 0
#else
#define EXA_ABL_COND_SLAB
#endif

// 8-byte LDS load that the back-end cannot pair into ds_read2_b64 (half the LDS rate of two ds_read_b64)
#define EXA_SLD(i) (*(const volatile __attribute__((address_space(3))) double*)(&lds[i]))

namespace exa {

// (opaque_v / opaque_s, exa_dg_kernels.hpp: values derived from them cannot be hoisted out of the step loop -- the compiler otherwise
// precomputes every address and predicate of every phase once per kernel and spills them, 324 VGPRs)

template <int N> struct StreamGeo {
    static constexpr int NN = N * N * N, NF = N * N;
    static constexpr int PY = (N % 2 == 0) ? N + 1 : N;      // row stride
    static constexpr int PX = N * PY;                         // plane stride
    static constexpr int SL = N * PX;                         // one (var, level) image
    static constexpr bool PERMUTE = (N == 8);                 // see the owner permutation in the kernel
    __device__ static inline int node_off(int n) { return (n / NF) * PX + ((n / N) % N) * PY + n % N; }
    __device__ static constexpr int pstride(int d) { return d == 0 ? PX : (d == 1 ? PY : 1); }
    // first node of pencil t (t = lexicographic index of the remaining axes = face-node index of the traces)
    __device__ static inline int pbase(int d, int t) {
        const int a = t / N, b = t - a * N;
        return d == 0 ? a * PY + b : (d == 1 ? a * PX + b : a * PX + b * PY);
    }
};

// HS = tasks per pencil in the derive phase (2: row halves, 40 VGPRs of partial sums; 1: whole pencils, 80),
// OH = owners per node in the load / fold phases (2: each owns half of the output levels; 1: all of them).
// HS = OH = 2 runs 16 waves at <= 128 VGPRs, HS = OH = 1 runs 8 waves at <= 256 VGPRs and reads each pencil once.
// The library instantiates HS = OH = 1 (dg_inst.hip); the other values were timed (profiles/r01_stream_kernel.txt) and
// are outside the parity tests.
template <int N, class PDE, int HS, int OH> struct StageAStream {
    using G = StreamGeo<N>;
    static constexpr int NV = PDE::NV, NA = PDE::NAUX;
    static constexpr int LG = 2;                              // time levels (slots) per step
    static constexpr int LS = (N + 1) / 2;                    // steps per iteration; slot ls of step g holds level ls*LS + g
    static constexpr int HR = (N + HS - 1) / HS;              // rows of a derive task
    static constexpr int LH = OH == 2 ? LS : N;               // output levels of a node owner
    static constexpr int SLOTS = LG / OH;                     // level slots a loader lane serves
    static constexpr int QSZ = NV * LG * G::SL;               // Q, A, B
    static constexpr int AXO = 3 * QSZ;                       // flux scalars [a][slot][node]
    static constexpr int PIC_D = 3 * QSZ + NA * LG * G::SL;   // Picard phases
    static constexpr int FIN_D = (pde_has_source<PDE>::value ? 5 : 4) * NV * G::SL;   // final phases: qbar, Fbar_x, Fbar_y, Fbar_z (, time-averaged source)
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)(PIC_D > FIN_D ? PIC_D : FIN_D);
    static constexpr int TH = LG * G::NF;                     // pencils per direction and step
    static constexpr int HW = ((TH + 63) / 64) * 64;          // lanes of one task part of a direction group (wave-aligned)
    static constexpr int GW = HS * HW;                        // lanes of a direction group
    static constexpr int OWNH = ((G::NN + 63) / 64) * 64;     // lanes of an owner part (wave-aligned)
    static constexpr int NT = (3 * GW > OH * OWNH) ? 3 * GW : OH * OWNH;
    static constexpr int SKEW_SLEEP = 32;                     // x 64 cycles x 32 slots ~ one iteration
    static constexpr size_t SLAB_D = (size_t)NV * N * G::NN;  // doubles of slab per workgroup: q[level][var][owner slot]
    // Levels per slot whose new iterate stays in registers over the iteration boundary (the others wait in the slab).
    // 1: 6 of 8 levels in the slab (120 KiB per workgroup, 3.84 MB per XCD: the cyclic sweep just misses the 4 MB L2 --
    // 47 GB of L2 <-> fabric traffic per 32^3-cell launch against 2.35 GB algorithmic); 2: 4 levels (80 KiB, 2.56 MB per XCD).
#ifndef EXA_STREAM_KEEP
#define EXA_STREAM_KEEP 2
#endif
    static constexpr int KEEP = (OH == 1 && LS >= 2) ? EXA_STREAM_KEEP : 1;
    static_assert(HS == 1 || HS == 2, "HS");
    static_assert(OH == 1 || OH == 2, "OH");
};

template <int N, class PDE, int HS, int OH>
__global__ void __launch_bounds__((StageAStream<N, PDE, HS, OH>::NT))
dg_stage_a_stream_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, double* __restrict__ trace,
                         long ncells, CellBox box, double dt, double idx0, double idx1, double idx2, int n_it,
                         const void* __restrict__ ops_raw, double* __restrict__ slab) {
    using G = StreamGeo<N>;
    using SA = StageAStream<N, PDE, HS, OH>;
    constexpr int NV = PDE::NV, NA = PDE::NAUX, DIM = 3;
    constexpr int NN = G::NN, NF = G::NF, SL = G::SL;
    constexpr int LG = SA::LG, LS = SA::LS, HR = SA::HR, LH = SA::LH, SLOTS = SA::SLOTS, QSZ = SA::QSZ, AXO = SA::AXO;
    constexpr int NT = SA::NT, TH = SA::TH, HW = SA::HW, GW = SA::GW, OWNH = SA::OWNH, KEEP = SA::KEEP;
    extern __shared__ __attribute__((aligned(16))) double lds[];

    const int tid = threadIdx.x;
    EXA_STAMP_INIT();
    const double idx[3] = {idx0, idx1, idx2};
    double* qs = slab + (size_t)blockIdx.x * SA::SLAB_D;

    // derive role: direction group, task part (rows of D), level slot, pencil
    const int grp = __builtin_amdgcn_readfirstlane(tid / GW);
    const int bt = tid - grp * GW;
    const int d_half = __builtin_amdgcn_readfirstlane(bt / HW);             // wave-uniform: selects the rows of D
    const int d_r = bt - d_half * HW;
    const int d_ls = d_r / NF, d_t = d_r - d_ls * NF;
    const bool d_task = grp < DIM && d_r < TH;
    // owner role: node x part of the output levels; loader role: node x level slot(s) (same split)
    const int o_h = __builtin_amdgcn_readfirstlane(tid / OWNH);
    const int o_slot = tid - o_h * OWNH;                                    // lane-linear: indexes the slab
    const bool owner = o_h < OH && o_slot < NN;
    int o_n = o_slot < NN ? o_slot : 0;                                     // node of this owner
    if constexpr (G::PERMUTE) {
        // Owner slot -> node permutation that makes the node-linear LDS phases (load, fold, averages) conflict-free
        // under the padded strides: in a block of 256 nodes (4 planes) every residue of node_off mod 32 occurs
        // exactly 8 times, so the k-th node of residue r goes to lane r of the k-th 32-lane group -- ds_read_b64
        // groups see 32 distinct double-banks, ds_write_b64 groups (16 lanes) 16 distinct ones.
        int* table = reinterpret_cast<int*>(lds);
        if (tid < NN) {
            const int n = tid, blk = n & ~255;
            const int r = G::node_off(n) & 31;
            int rank = 0;
            for (int m = blk; m < n; m++) rank += ((G::node_off(m) & 31) == r) ? 1 : 0;
            table[blk + rank * 32 + r] = n;
        }
        __syncthreads();
        o_n = table[o_slot < NN ? o_slot : 0];
        __syncthreads();
    }
    const int o_off0 = G::node_off(o_n);
    // level of owned accumulator k:  OH == 2: o_h*LS + k (slot o_h, step k);  OH == 1: k (slot k / LS, step k % LS)

    // Start the workgroups of an XCD staggered over roughly one Picard iteration: they all do the same work, so
    // without it every CU writes its share of the slab in the same microsecond (31 MB chip-wide at N = 8).
    for (int k = (blockIdx.x >> 3) & 31; k > 0; k--) __builtin_amdgcn_s_sleep(SA::SKEW_SLEEP);

    for (long b = blockIdx.x; b < box.nbox; b += gridDim.x) {
        const long cell = box.cell(b);
        double acc[LH][NV];                                      // time-update accumulators, then the final iterate
        double nxt[SLOTS][NV];                                   // what the next load phase needs, fetched a step ahead
        double kept[KEEP > 1 ? SLOTS : 1][NV];                   // KEEP == 2: the second level of each slot (never in the slab)
#pragma unroll
        for (int si = 0; si < (KEEP > 1 ? SLOTS : 1); si++)
#pragma unroll
            for (int v = 0; v < NV; v++) kept[si][v] = 0.0;
#pragma unroll
        for (int k = 0; k < LH; k++)
#pragma unroll
            for (int v = 0; v < NV; v++) acc[k][v] = 0.0;
#pragma unroll
        for (int si = 0; si < SLOTS; si++)
#pragma unroll
            for (int v = 0; v < NV; v++) nxt[si][v] = (owner && si == 0) ? u_in[(cell * NN + o_n) * NV + v] : 1.0;

        for (int it = 0; it < n_it; it++) {
            // iteration 0: the iterate is constant in time -- one level, row sums of T.  Later iterations: step g
            // takes the levels g and LS + g, so that every lane loads what it wrote itself (no fence), a step ahead
            // of its use (the slab latency hides under a whole step).
            const int ngroups = it == 0 ? 1 : LS;
            for (int g = 0; g < ngroups; g++) {
                const int nl = it == 0 ? 1 : (LS + g < N ? 2 : 1);
                // ---- load: q of the level(s), flux scalars once per node; then fetch ahead
                if (owner) {
                    const int base = opaque_v(o_off0);
#pragma unroll
                    for (int si = 0; si < SLOTS; si++) {
                        const int ls = OH == 2 ? opaque_s(o_h) : si;
                        if (ls < nl EXA_ABL_COND_LOAD) {
                            double a[nz(NA)];
                            const int o_off = base + ls * SL;
                            PDE::aux_fast(nxt[si], a);
#pragma unroll
                            for (int v = 0; v < NV; v++) lds[v * LG * SL + o_off] = nxt[si][v];
#pragma unroll
                            for (int k = 0; k < NA; k++) lds[AXO + k * LG * SL + o_off] = a[k];
                        }
                    }
                    if (g + 1 < ngroups) {                       // the next level of each slot, from the slab
                        const int sl = opaque_v(o_slot);
#pragma unroll
                        for (int si = 0; si < SLOTS; si++) {
                            const int ls = OH == 2 ? opaque_s(o_h) : si;
                            const int ln = ls * LS + g + 1;
                            if (KEEP > 1 && g == 0) {                            // level 1 of the slot: still in registers
#pragma unroll
                                for (int v = 0; v < NV; v++) nxt[si][v] = kept[KEEP > 1 ? si : 0][v];
                            } else
                            if (ln < N EXA_ABL_COND_SLAB) {
                                const double* row = qs + (size_t)ln * NV * NN;   // uniform base + lane offset: saddr loads
#pragma unroll
                                for (int v = 0; v < NV; v++) nxt[si][v] = row[v * NN + sl];
                            }
                        }
                    } else {                                     // last step: u for the update
                        const double* un = u_in + (cell * NN + opaque_v(o_n)) * NV;
#pragma unroll
                        for (int v = 0; v < NV; v++) nxt[0][v] = un[v];
                    }
                }
                EXA_STAMP(0);
                __syncthreads();
                EXA_STAMP(1);
                // ---- derive: task = rows [half*HR, half*HR + HR) of one pencil, all variables
                double s[HR][NV];
                bool hold = false;
                int hoff = 0;
                static_for<0, DIM>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    if (d_task && grp == D && d_ls < nl EXA_ABL_COND_D) {
                        constexpr int ps = G::pstride(D);
                        const int half = HS == 1 ? 0 : opaque_s(d_half);
                        const EXA_AS4 double* Dcol = ops_here<N>(ops_raw)->DT + half * HR;     // DT[j][half*HR + i]
                        const int off = opaque_v(d_ls * SL + G::pbase(D, d_t));
                        const int soff = off + half * (HR * ps);                               // first row of this task
                        if constexpr (HS == 1) {
                            // even-odd form of the centro-antisymmetric D (DgOps::DEO): half the FMAs; odd N: the middle
                            // node and the middle row on top
                            constexpr int H = N / 2;
                            const EXA_AS4 double* Em = ops_here<N>(ops_raw)->DEO;
                            double P[H][NV], M[H][NV], mid[NV];
#pragma unroll
                            for (int i = 0; i < H; i++)
#pragma unroll
                                for (int v = 0; v < NV; v++) P[i][v] = M[i][v] = 0.0;
#pragma unroll
                            for (int v = 0; v < NV; v++) mid[v] = 0.0;
#pragma unroll
                            for (int j = 0; j < H; j++) {
                                double qa[NV], aa[nz(NA)], Fa[NV], qb[NV], ab[nz(NA)], Fb[NV];
#pragma unroll
                                for (int v = 0; v < NV; v++) {
                                    qa[v] = EXA_SLD(off + v * LG * SL + j * ps);
                                    qb[v] = EXA_SLD(off + v * LG * SL + (N - 1 - j) * ps);
                                }
#pragma unroll
                                for (int k = 0; k < NA; k++) {
                                    aa[k] = EXA_SLD(AXO + off + k * LG * SL + j * ps);
                                    ab[k] = EXA_SLD(AXO + off + k * LG * SL + (N - 1 - j) * ps);
                                }
                                PDE::template flux_scaled<D>(qa, aa, idx[D], Fa);
                                PDE::template flux_scaled<D>(qb, ab, idx[D], Fb);
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
                            if constexpr (N % 2 == 1) {          // middle node
                                double qa[NV], aa[nz(NA)], Fa[NV];
#pragma unroll
                                for (int v = 0; v < NV; v++) qa[v] = EXA_SLD(off + v * LG * SL + H * ps);
#pragma unroll
                                for (int k = 0; k < NA; k++) aa[k] = EXA_SLD(AXO + off + k * LG * SL + H * ps);
                                PDE::template flux_scaled<D>(qa, aa, idx[D], Fa);
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
                        } else {
#pragma unroll
                        for (int i = 0; i < HR; i++)
#pragma unroll
                            for (int v = 0; v < NV; v++) s[i][v] = 0.0;
#pragma unroll
                        for (int j = 0; j < N; j++) {
                            double q[NV], a[nz(NA)], F[NV];
#pragma unroll
                            for (int v = 0; v < NV; v++) q[v] = EXA_SLD(off + v * LG * SL + j * ps);
#pragma unroll
                            for (int k = 0; k < NA; k++) a[k] = EXA_SLD(AXO + off + k * LG * SL + j * ps);
                            PDE::template flux_scaled<D>(q, a, idx[D], F);
#pragma unroll
                            for (int i = 0; i < HR; i++) {
                                const double dij = Dcol[j * N + i];
#pragma unroll
                                for (int v = 0; v < NV; v++) s[i][v] += dij * F[v];
                            }
                        }
                        }
                        if constexpr (D > 0) {
#pragma unroll
                            for (int i = 0; i < HR; i++)
                                if (HS * HR == N || i + 1 < HR || half == 0) {                 // odd N: the second half has one row less
#pragma unroll
                                    for (int v = 0; v < NV; v++) lds[D * QSZ + soff + v * LG * SL + i * ps] = s[i][v];
                                }
                        } else {
                            if constexpr (pde_has_source<PDE>::value) {          // q_t + div F = S(q): the x pencils carry -S(q) of their nodes
                                static_assert(HS == 1 || !pde_has_source<PDE>::value, "source terms: whole-pencil derive tasks only");
#pragma unroll
                                for (int i = 0; i < HR; i++) {
                                    double qi[NV], Sq[NV];
#pragma unroll
                                    for (int v = 0; v < NV; v++) qi[v] = EXA_SLD(off + v * LG * SL + i * ps);     // (Q is intact until the barrier)
                                    PDE::source(qi, Sq);
#pragma unroll
                                    for (int v = 0; v < NV; v++) s[i][v] -= Sq[v];
                                }
                            }
                            hold = true;
                            hoff = soff;
                        }
                    }
                });
                EXA_STAMP(2);
                __syncthreads();
                EXA_STAMP(3);
                if (hold) {                                      // every read of Q is done: Q := S_x
                    constexpr int ps = G::pstride(0);
#pragma unroll
                    for (int i = 0; i < HR; i++)
                        if (HS * HR == N || i + 1 < HR || d_half == 0) {
#pragma unroll
                            for (int v = 0; v < NV; v++) lds[hoff + v * LG * SL + i * ps] = s[i][v];
                        }
                }
                EXA_STAMP(4);
                __syncthreads();
                EXA_STAMP(5);
                // ---- fold: acc[l'] += T[l'][l] * S_l for the levels of this step
                if (owner EXA_ABL_COND_FOLD) {
                    // T and Tsum are adjacent in DgOps: one scalar-indexed array serves both (a select between two
                    // pointers turns the coefficient into a serialised vector load)
                    const EXA_AS4 double* Tm = ops_here<N>(ops_raw)->T;
                    const int lp0 = OH == 2 ? opaque_s(o_h) * LS : 0;
                    const int o_off = opaque_v(o_off0);
                    if (g == 0) {
#pragma unroll
                        for (int k = 0; k < LH; k++)
#pragma unroll
                            for (int v = 0; v < NV; v++) acc[k][v] = 0.0;
                    }
#pragma unroll
                    for (int ls = 0; ls < LG; ls++) {
                        if (ls < nl) {
                            const int l = ls * LS + g;
                            double S[NV], tl[LH];
#pragma unroll
                            for (int k = 0; k < LH; k++) {
                                const int lp = (LH * OH == N || k + 1 < LH || lp0 + k < N) ? lp0 + k : 0;
                                tl[k] = Tm[__builtin_amdgcn_readfirstlane(it == 0 ? N * N + lp : lp * N + l)];
                            }
                            double Sx[NV], Sy[NV], Sz[NV];
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                const int p = (v * LG + ls) * SL + o_off;
                                Sx[v] = EXA_SLD(p);
                                Sy[v] = EXA_SLD(p + QSZ);
                                Sz[v] = EXA_SLD(p + 2 * QSZ);
                            }
#pragma unroll
                            for (int v = 0; v < NV; v++) S[v] = Sx[v] + Sy[v] + Sz[v];
#pragma unroll
                            for (int k = 0; k < LH; k++)
#pragma unroll
                                for (int v = 0; v < NV; v++) acc[k][v] += tl[k] * S[v];
                        }
                    }
                }
                EXA_STAMP(6);
#ifdef EXA_STREAM_BARRIER4
                __syncthreads();
#else
                static_assert(OH == 1, "two owners per node: the fold reads the co-owner's slot -- build with -DEXA_STREAM_BARRIER4");
#endif
                // (no barrier here: the next load phase writes Q and the flux scalars at the owner's OWN node, which only this lane
                //  read in the fold above; the derive phase that read them across lanes ended two barriers ago)
                EXA_STAMP(7);
            }
            // ---- new iterate q_l' = u - dt * acc (u arrived in nxt[0]); the first level of each slot feeds the next
            // load phase from registers, the others wait in the slab
            if (owner) {
                const int lp0 = OH == 2 ? opaque_s(o_h) * LS : 0;
                const int n = opaque_v(o_slot);
                const bool keep = it + 1 < n_it;
#pragma unroll
                for (int k = 0; k < LH; k++) {
#pragma unroll
                    for (int v = 0; v < NV; v++) acc[k][v] = nxt[0][v] - dt * acc[k][v];
                    const bool first = OH == 2 ? k == 0 : (k % LS < KEEP);         // stays in registers (nxt / kept)
                    if (!first && keep EXA_ABL_COND_SLAB && (LH * OH == N || k + 1 < LH || lp0 + k < N)) {
                        double* row = qs + (size_t)(lp0 + k) * NV * NN;          // uniform base + lane offset: saddr stores
#pragma unroll
                        for (int v = 0; v < NV; v++) row[v * NN + n] = acc[k][v];
                    }
                }
#pragma unroll
                for (int si = 0; si < SLOTS; si++)
#pragma unroll
                    for (int v = 0; v < NV; v++) nxt[si][v] = acc[OH == 2 ? 0 : si * LS][v];
                if constexpr (KEEP > 1) {
#pragma unroll
                    for (int si = 0; si < SLOTS; si++)
#pragma unroll
                        for (int v = 0; v < NV; v++) kept[si][v] = (si * LS + 1 < N) ? acc[si * LS + 1 < LH ? si * LS + 1 : 0][v] : 0.0;
                }
            }
            EXA_STAMP(8);
        }

        // ---- time averages: each owner over its levels (OH == 2: the two parts meet in LDS): qbar | Fbar_d
#ifndef EXA_STREAM_BARRIER4
        __syncthreads();                                         // the last fold of every lane is done: the final image reuses Q / A / B
#endif
        {
            double qb[NV], Fb[DIM][NV];
            [[maybe_unused]] double Sbar[NV];
            if constexpr (pde_has_source<PDE>::value) {
#pragma unroll
                for (int v = 0; v < NV; v++) Sbar[v] = 0.0;
            }
            const int o_off = opaque_v(o_off0);
            if (owner) {
                const EXA_AS4 double* wm = ops_here<N>(ops_raw)->w;
                const int lp0 = OH == 2 ? o_h * LS : 0;
#pragma unroll
                for (int v = 0; v < NV; v++) qb[v] = 0.0;
#pragma unroll
                for (int d = 0; d < DIM; d++)
#pragma unroll
                    for (int v = 0; v < NV; v++) Fb[d][v] = 0.0;
                if (n_it > 0) {
#pragma unroll
                    for (int k = 0; k < LH; k++) {
                        const int lp = lp0 + k;
                        if (lp < N) {
                            double a[nz(NA)], F[NV];
                            PDE::aux_fast(acc[k], a);
                            const double wl = wm[lp];
#pragma unroll
                            for (int v = 0; v < NV; v++) qb[v] += wl * acc[k][v];
                            static_for<0, DIM>([&](auto dc) {
                                constexpr int D = decltype(dc)::value;
                                PDE::template flux<D>(acc[k], a, F);
#pragma unroll
                                for (int v = 0; v < NV; v++) Fb[D][v] += wl * F[v];
                            });
                            if constexpr (pde_has_source<PDE>::value) {
                                double Sq[NV];
                                PDE::source(acc[k], Sq);
#pragma unroll
                                for (int v = 0; v < NV; v++) Sbar[v] += wl * Sq[v];
                            }
                        }
                    }
                } else if (o_h == 0) {                           // single stage: qbar = u, Fbar = F(u)
                    double a[nz(NA)];
                    PDE::aux_fast(nxt[0], a);
#pragma unroll
                    for (int v = 0; v < NV; v++) qb[v] = nxt[0][v];
                    static_for<0, DIM>([&](auto dc) {
                        constexpr int D = decltype(dc)::value;
                        PDE::template flux<D>(nxt[0], a, Fb[D]);
                    });
                    if constexpr (pde_has_source<PDE>::value) PDE::source(nxt[0], Sbar);
                }
                if (o_h == OH - 1) {
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        lds[v * SL + o_off] = qb[v];
#pragma unroll
                        for (int d = 0; d < DIM; d++) lds[((1 + d) * NV + v) * SL + o_off] = Fb[d][v];
                        if constexpr (pde_has_source<PDE>::value) {
                            static_assert(OH == 1 || !pde_has_source<PDE>::value, "source terms: one owner per node only");
                            lds[(4 * NV + v) * SL + o_off] = Sbar[v];
                        }
                    }
                }
            }
            __syncthreads();
            if constexpr (OH == 2) {
                if (owner && o_h == 0) {
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        lds[v * SL + o_off] += qb[v];
#pragma unroll
                        for (int d = 0; d < DIM; d++) lds[((1 + d) * NV + v) * SL + o_off] += Fb[d][v];
                    }
                }
                __syncthreads();
            }
        }
        EXA_STAMP(9);

        // ---- volume integral (in place over Fbar_d) + face extrapolation: pencil tasks (d, v, t), t fastest
        {
            const EXA_AS4 DgOps<N>* o = ops_here<N>(ops_raw);
            for (int task = tid; task < DIM * NV * NF; task += NT) {
                const int d = task / (NV * NF);
                const int r = task - d * (NV * NF);
                const int v = r / NF, t = r - v * NF;
                const int ps = G::pstride(d);
                const int pb = G::pbase(d, t);
                double qb[N], Fb[N];
#pragma unroll
                for (int j = 0; j < N; j++) {
                    qb[j] = EXA_SLD(v * SL + pb + j * ps);
                    Fb[j] = EXA_SLD(((1 + d) * NV + v) * SL + pb + j * ps);
                }
                const double sc = dt * idx[d];
#pragma unroll
                for (int i = 0; i < N; i++) {
                    double sv = 0.0;
#pragma unroll
                    for (int j = 0; j < N; j++) sv += o->Kxi[i * N + j] * Fb[j];
                    lds[((1 + d) * NV + v) * SL + pb + i * ps] = sc * o->iw[i] * sv;
                }
                double qL = 0.0, qR = 0.0, FL = 0.0, FR = 0.0;
#pragma unroll
                for (int j = 0; j < N; j++) {
                    qL += o->phiL[j] * qb[j];
                    qR += o->phiR[j] * qb[j];
                    FL += o->phiL[j] * Fb[j];
                    FR += o->phiR[j] * Fb[j];
                }
                double* tl = trace + (((long)d * 2 + 0) * ncells + cell) * (2 * NV * NF);
                double* tr = trace + (((long)d * 2 + 1) * ncells + cell) * (2 * NV * NF);
                tl[(0 * NV + v) * NF + t] = qL;
                tl[(1 * NV + v) * NF + t] = FL;
                tr[(0 * NV + v) * NF + t] = qR;
                tr[(1 * NV + v) * NF + t] = FR;
            }
        }
        __syncthreads();
        EXA_STAMP(10);

        // ---- u* = u + sum_d vol_d, AoS (coalesced)
        for (int e = tid; e < NN * NV; e += NT) {
            const int n = e / NV, v = e - n * NV;
            const int off = G::node_off(n);
            double us = u_in[cell * (NN * NV) + e];
            if constexpr (pde_has_source<PDE>::value) us += dt * lds[(4 * NV + v) * SL + off];
#pragma unroll
            for (int d = 0; d < DIM; d++) us += lds[((1 + d) * NV + v) * SL + off];
            u_out[cell * (NN * NV) + e] = us;
        }
        __syncthreads();                                         // LDS is reused by the next cell
        EXA_STAMP(11);
    }
    EXA_STAMP_FLUSH();
}

}  // namespace exa
