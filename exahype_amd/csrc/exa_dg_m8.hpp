// Stage A for 3-D cells with N = 8 (p = 7, BASELINE configs[4]) with the derivative contraction on the MATRIX pipe and the whole
// space-time iterate in registers -- nothing outside the chip (the level-streamed kernel of exa_dg_stream.hpp parks half of the
// iterate in a lane-private slab in HBM / L2: 170 GB of L2 <-> fabric traffic per 64^3 launch against 18.8 GB algorithmic).  Same
// scheme and results (to rounding) as dg_stage_a_stream_kernel<8>; no counterpart in the reference.
//
// The even-odd form of the centro-antisymmetric 8 x 8 derivative operator is two dense 4 x 4 blocks: with e_j = F_j + F_{7-j},
// o_j = F_j - F_{7-j} (j < 4),  P = Ee e,  M = Eo o,  s_i = M_i + P_i,  s_{7-i} = M_i - P_i.  v_mfma_f64_4x4x4_4b_f64 multiplies four
// independent 4 x 4 blocks per instruction with the operands spread over the lanes (A[i][k] in lane i + 4 blk + 16 k, B[k][c] in
// lane c + 4 blk + 16 k, D[i][c] in lane c + 4 blk + 16 i: probed by scripts/mfma_n8_probe.hip), so FOUR lanes share a pencil:
// lane (p = lane & 15, j = lane >> 4) loads the node pair (j, 7 - j) of pencil p, evaluates its two fluxes, hands e_j (o_j) to the
// instruction as its B element and receives P_j (M_j) of ITS OWN pencil -- the sums of the nodes it loaded.  No cross-lane movement,
// 100 % tile fill, the even-odd halving kept, bit-identical to the vector form (same FMA order).  Per pencil and variable 2 MFMAs
// of 256 multiply-adds replace 32 vector FMAs; a derive task needs ~60 VGPRs instead of ~130 (no 4 x 5 x 2 accumulator block),
// which is what lets the owner lanes keep all 8 levels of the iterate AND the 8 accumulators of the time contraction (160 VGPRs).
// Measured beside the vector form on LDS-resident data: scripts/mfma_n8.hip, profiles/r03_mfma_n8.txt.
//
// Workgroup = 512 threads = 512 node owners, one per CU (156.7 KB of LDS), persistent grid.  A step takes two time levels:
//   load    owners: q_l and the flux scalars of the two levels -> LDS (issued inside the previous fold)
//   derive  three rounds (x, y, z), in each all 8 waves: wave w = pencils 16 w .. 16 w + 15 of the 128 (level slot, pencil) pairs;
//           y -> A, z -> B, x stays in registers (10 values) and overwrites Q after the barrier
//   fold    owners: acc[l'] += -dt T[l'][l] (S_x + S_y + S_z)_l, started from u: after the last step acc is the new iterate
// LDS image and strides as exa_dg_stream.hpp (StreamGeo<8>: row stride 9, plane stride 72; Q | A | B | flux scalars).
#pragma once
#include "exa_dg_stream.hpp"

#ifndef EXA_M8_XINPLACE
#define EXA_M8_XINPLACE 1
#endif
#ifndef EXA_M8_LAYOUT          // 1: r5 placement of the sums (conflict-free stores for y and z, see M8Geo below); 0: all three in the layout of Q (r3 / r4)
#define EXA_M8_LAYOUT 0
#endif
#ifndef EXA_M8_PLANE           // 1: r5, a wave OWNS a plane of the cell (fixed y index): its x and z pencils are its own, two of the three barriers of a step go (M8Geo below)
#define EXA_M8_PLANE 0
#endif
#if (EXA_M8_LAYOUT || EXA_M8_PLANE) && !EXA_M8_XINPLACE
#error "EXA_M8_LAYOUT / EXA_M8_PLANE store one direction in place: they need EXA_M8_XINPLACE=1"
#endif
#if EXA_M8_LAYOUT && EXA_M8_PLANE
#error "EXA_M8_LAYOUT and EXA_M8_PLANE are alternatives"
#endif
// order between LDS accesses of different lanes of ONE wave (the hardware completes a wave's LDS operations in order; this keeps the compiler from moving them)
#define EXA_M8_WAVE_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#if EXA_M8_PLANE
#define EXA_M8_SYNC_PUTS() do { } while (0)      // the first round of a step reads what the wave itself put
#else
#define EXA_M8_SYNC_PUTS() __syncthreads()
#endif
#ifdef EXA_M8_ROUND_STAMPS
#define EXA_RSTAMP(k) EXA_STAMP(k)
#else
#define EXA_RSTAMP(k) do { } while (0)
#endif

namespace exa {

// LDS strides: row stride 8, plane stride 70, level stride 560 -- with them the lane tables (dg_inst.hip fill_m8_tables) make every LDS
// read of the two-level derive rounds conflict-free in all three directions.  A 32-lane ds_read_b64 group holds the node pairs k and
// k + 1 of 16 pencils, so the pencils of a wave need bank residues R with R and R + stride disjoint: for a pencil stride s the
// residues are 2-coloured along the cycles of r -> r + s (mod 32), four waves take one colour each.  (Strides 9 / 72 of
// exa_dg_stream.hpp leave the x pencils 2- to 3-way conflicted in this lane mapping; found by search over the strides that fit.)
struct M8Geo {
    static constexpr int N = 8, NN = 512, NF = 64;
#if EXA_M8_PLANE
    // r5, plane ownership: wave w of the workgroup owns the 64 nodes (a, b = w, c) of the cell -- every x pencil (b, c) and z pencil (a, b) with b = w lies in
    // its own nodes, so only the y rounds exchange data between waves.  Strides 8 / 68 / 562: the owners' nodes 4 a + c cover every bank residue exactly twice
    // (lane = 32 (c >> 2) + ((4 a + c) & 31): conflict-free puts and sum reads); a z round (16 pencils = 8 a x 2 level slots, bases 4 a + 18 ls mod 32 = the 16
    // even residues) reads conflict-free; an x round (8 c x 2 level slots, bases c + 18 ls, lane rows in the order 0, 2, 1, 3 of the node pairs: distance 2 x 68 = 8
    // mod 32) keeps a 2-way conflict on 2 of 32 lanes; the y rounds take their lanes from the 2-colouring search of dg_inst.hip as before.
    static constexpr int PY = 8, PX = 68, SL = 562;
#else
    static constexpr int PY = 8, PX = 70, SL = 560;
#endif
    __host__ __device__ static inline int node_off(int n) { return (n / NF) * PX + ((n / N) % N) * PY + n % N; }
    __host__ __device__ static constexpr int pstride(int d) { return d == 0 ? PX : (d == 1 ? PY : 1); }
    __host__ __device__ static inline int pbase(int d, int t) {
        const int a = t / N, b = t - a * N;
        return d == 0 ? a * PY + b : (d == 1 ? a * PX + b : a * PX + b * PY);
    }
    // r5 (EXA_M8_LAYOUT): where the sums go.  A 16-lane ds_write_b64 group is 16 pencils at ONE node, so a store is conflict-free iff the pencils' bases differ
    // mod 16 doubles; a 32-lane ds_read_b64 group is 16 pencils at the nodes of two lane rows k, k + 1, conflict-free iff R and R + (node distance) * stride are
    // disjoint mod 32 (R = the bases).  In the layout of Q both hold together only if (node distance) * stride = 16 mod 32:
    //   y (stride 8): the lane rows take the node pairs in the order 0, 2, 1, 3 (distance 2: the matrix instruction sums over its k index, any order of the
    //      pairs will do if the operator entries follow), a wave = the pencils (a, c) with a in {k, k + 4} -- sums IN PLACE over Q, after the barrier;
    //   z (stride 1): impossible in the layout of Q -- its sums go to a compact array [variable][level slot][wave w'][node c][lane p] (16 consecutive doubles
    //      per store group); a wave = the 4 x 4 block of pencils (a, b) with a >> 2, b >> 2 fixed, p = 4 (a & 3) + (b & 3);
    //   x (stride 70): array A in the layout of Q as before (its store groups stay 2-way conflicted).
    // The node owners read all three at their node: a 32-lane owner group = the nodes (a, b, c) of one 4 x 4 block of (a, b) and one pair c in {2 m, 2 m + 1} --
    // 32 distinct bank residues in Q (6 a + 8 b runs over the 16 even residues, c adds 0 / 1) and 32 distinct ones in the z array (16 (c & 1) + p).
    static constexpr int ZV = 2 * NN, ZL = NN;                        // z array: variable stride, level-slot stride
    __host__ __device__ static constexpr int ypair(int j) { return ((j & 1) << 1) | (j >> 1); }      // lane row j -> node pair (0, 2, 1, 3)
    static constexpr int PERM_D = EXA_M8_LAYOUT ? 1 : (EXA_M8_PLANE ? 0 : -1);                         // the direction whose rounds take the pairs in that order
    __host__ __device__ static inline int plane_slot(int n) {        // EXA_M8_PLANE: node -> owner thread (wave = y index, lane by bank residue)
        const int a = n / NF, b = (n / N) % N, c = n % N;
        return b * 64 + (c >> 2) * 32 + ((4 * a + c) & 31);
    }
    __host__ __device__ static inline int zslot(int n) {             // node -> offset in a (variable, level slot) block of the z array
        const int a = n / NF, b = (n / N) % N, c = n % N;
        return (2 * (a >> 2) + (b >> 2)) * 128 + c * 16 + 4 * (a & 3) + (b & 3);
    }
    __host__ __device__ static inline int owner_slot(int n) {        // node -> owner lane (group * 32 + bank residue in Q)
        const int a = n / NF, b = (n / N) % N, c = n % N;
        return ((((a >> 2) << 3) | ((b >> 2) << 2) | (c >> 1)) << 5) | ((6 * a + 8 * b + c) & 31);
    }
};

template <class PDE> struct StageAM8 {
    static constexpr int N = 8;
    using G = M8Geo;
    static constexpr int NV = PDE::NV, NA = PDE::NAUX;
    static constexpr int NT = 512, LG = 2, LS = N / LG;
    static constexpr int VS = LG * G::SL;                             // slot stride
    static constexpr int QSZ = NV * VS;                               // Q, A, B
    static constexpr int AXO = 3 * QSZ;                               // flux scalars
    static constexpr int PIC_D = 3 * QSZ + NA * VS;
    static constexpr int FIN_D = (pde_has_source<PDE>::value ? 5 : 4) * NV * G::SL;
    static constexpr int IMG_D = PIC_D > FIN_D ? PIC_D : FIN_D;
    // behind the cell's image: the lane tables as 16-bit offsets [table][direction][lane] (0xffff = no pencil) -- read in front of every
    // round instead of being held in (and spilled from) registers through the Picard loop
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)IMG_D + sizeof(unsigned short) * 2 * 3 * NT;
    static constexpr bool FITS = LDS_BYTES <= 160 * 1024 && G::NN == NT;
    // lane tables behind the operator image: [levels in the step: 2, 1][direction][lane] -> level slot * SL + first node of the lane's pencil, -1 idle
    static constexpr int TAB_INTS = 2 * 3 * NT;
};

template <class PDE>
__global__ void __launch_bounds__(512)
dg_stage_a_m8_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, double* __restrict__ trace,
                     long ncells, CellBox box, double dt, double idx0, double idx1, double idx2, int n_it,
                     const void* __restrict__ ops_raw, const void* __restrict__ step_raw, const int* __restrict__ tab, PlainGeo geo) {
    constexpr int N = 8, H = 4, DIM = 3;
    // Term sets whose terms depend on position / time (XT) or carry a non-conservative product (NCP) -- generated term sets only (the hooks of
    // `Unit test/correctness_test.cpp:16-41,145-155`), as in exa_dg_reg.hpp: coordinates and level times reach the flux and the source, the lane's
    // two nodes get B_D(q) (D q) / h_D from two more matrix instructions per variable on q itself, the time-averaged ncp term is one more pass of
    // the derive rounds over the final iterate.  Under `if constexpr`: the built-in term sets compile to the same code as before.
    constexpr bool XT = pde_has_xt<PDE>::value, NCP = pde_has_ncp<PDE>::value;
    constexpr bool FXT = pde_flux_xt<PDE>::value;                     // the flux itself sees x, t (otherwise only the source / ncp do)
    using G = M8Geo;
    using SA = StageAM8<PDE>;
    constexpr int NV = SA::NV, NA = SA::NA, NN = G::NN, NF = G::NF, SL = G::SL, VS = SA::VS, QSZ = SA::QSZ, AXO = SA::AXO, NT = SA::NT, LS = SA::LS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    [[maybe_unused]] const int bt = lane, grp = wave & 3;             // (stamp builds)
    EXA_STAMP_INIT();
#ifndef EXA_M8_STAGGER
#define EXA_M8_STAGGER 1
#endif
    // the two waves of a SIMD (wave and wave + 4) at different static issue priorities: the favoured one runs ahead inside a phase, so that its
    // LDS traffic falls beside the other's arithmetic (12.23 -> 12.13 ms per 32^3 launch; level 1 or 3: the same)
    if (EXA_M8_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_setprio(EXA_M8_STAGGER);

    // owner slot -> node permutation that makes the node-linear LDS phases conflict-free under the padded strides (exa_dg_stream.hpp)
    int o_n;
    {
        int* table = reinterpret_cast<int*>(lds);
#if EXA_M8_LAYOUT
        table[G::owner_slot(tid)] = tid;                              // (a bijection)
#elif EXA_M8_PLANE
        table[G::plane_slot(tid)] = tid;                              // (a bijection)
#else
        const int n = tid, blk = n & ~255;
        const int r = G::node_off(n) & 31;
        int rank = 0;
        for (int m = blk; m < n; m++) rank += ((G::node_off(m) & 31) == r) ? 1 : 0;
        table[blk + rank * 32 + r] = n;
#endif
        __syncthreads();
        o_n = table[tid];
        __syncthreads();
    }
    const int o_off = G::node_off(o_n);
    [[maybe_unused]] const int o_z = G::zslot(o_n);
    // the three directional sums of level slot ls, variable v at the owner's node
    auto sums_at = [&](int ls, int v) -> double {
        const int p = o_off + ls * SL + v * VS;
#if EXA_M8_LAYOUT
        const double sy = EXA_SLD(p), sx = EXA_SLD(p + QSZ), sz = EXA_SLD(2 * QSZ + v * G::ZV + ls * G::ZL + o_z);
#else
        const double sx = EXA_SLD(p), sy = EXA_SLD(p + QSZ), sz = EXA_SLD(p + 2 * QSZ);
#endif
        return sx + (sy + sz);
    };
    [[maybe_unused]] double xc[3] = {0.0, 0.0, 0.0};                  // low corner of the current cell
    [[maybe_unused]] double xio[3] = {0.0, 0.0, 0.0};                 // reference coordinates of the owner's node
    if constexpr (XT) {
        xio[0] = xi_of<N>(geo, o_n / NF);
        xio[1] = xi_of<N>(geo, (o_n / N) % N);
        xio[2] = xi_of<N>(geo, o_n % N);
    }
    [[maybe_unused]] auto owner_x = [&](double (&x)[3]) {
#pragma unroll
        for (int a = 0; a < 3; a++) x[a] = xc[a] + xio[a] * geo.h[a];
    };
    [[maybe_unused]] auto level_t = [&](int l) -> double { return geo.t + geo.xi[l] * dt; };
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

    // derive role: node pair j = lane >> 4 of the pencil the lane tables give this lane (one per direction; the 16 pencils of a wave
    // have bank residues that keep the pairs k, k + 1 of a 32-lane group apart)
    const int d_j = lane >> 4;
    unsigned short* ltab = reinterpret_cast<unsigned short*>(lds + SA::IMG_D);
#pragma unroll
    for (int k = 0; k < 6; k++) ltab[k * NT + tid] = (unsigned short)tab[k * NT + tid];      // (-1 -> 0xffff)
    __syncthreads();
    // A operands of this lane: Ee[k][i], Eo[k][i] with i = lane & 3, k = lane >> 4 (DgOps::DEO packing: [j][i] even part, [j][H + i] odd part)
    const double aEe = static_cast<const DgOps<N>*>(ops_raw)->DEO[d_j * N + (lane & 3)];
    const double aEo = static_cast<const DgOps<N>*>(ops_raw)->DEO[d_j * N + H + (lane & 3)];
#if EXA_M8_LAYOUT || EXA_M8_PLANE     // rounds of direction M8Geo::PERM_D: rows and columns of the two 4 x 4 blocks in the order of the node pairs 0, 2, 1, 3
    const double aEeY = static_cast<const DgOps<N>*>(ops_raw)->DEO[G::ypair(d_j) * N + G::ypair(lane & 3)];
    const double aEoY = static_cast<const DgOps<N>*>(ops_raw)->DEO[G::ypair(d_j) * N + H + G::ypair(lane & 3)];
#else
    const double aEeY = aEe, aEoY = aEo;
#endif

    // one round: direction D, results to A (D = 1), B (D = 2) or into hx (D = 0)
    // tA / tB: times of the two level slots (XT); mode 0: the derivative sums, 1: only their non-conservative part (the closing pass)
    // stage (EXA_M8_XEARLY): 0 the whole round; 1 only its 14 operand loads, handed out through `ext`; 2 everything behind them, the operands taken from `ext`
    // -- the in-place round requests its operands in FRONT of the barrier it runs behind (they are in Q and the scalars, which nobody writes before it)
    struct RoundOps { double qa[NV], qb[NV], aa[nz(NA)], ab[nz(NA)]; int off; };
    auto round_staged = [&](auto dc, int tb, double (&hx)[2][NV], [[maybe_unused]] double tA, [[maybe_unused]] double tB, auto mode, auto stage, RoundOps& ext) {
        constexpr int D = decltype(dc)::value;
        constexpr int MODE = decltype(mode)::value;
        constexpr int ST = decltype(stage)::value;
        int off;
        if constexpr (ST == 2) off = ext.off;
        else off = ltab[(tb * 3 + D) * NT + opaque_v(tid)];
        if constexpr (ST == 1) ext.off = off;
        if (off != 0xffff) {                                          // (wave-uniform: a wave has 16 pencils or none)
            constexpr int ps = G::pstride(D);
            // (lane constants behind opaque copies: hoisted out of the cell loop their products with the strides live through the Picard
            // iterations and spill -- 96 B of scratch, reloaded in front of every round)
            const int dj0 = opaque_v(d_j);
            const int dj = D == G::PERM_D ? G::ypair(dj0) : dj0;                 // the node pair (dj, 7 - dj) of this lane row
            const double oEe = D == G::PERM_D ? aEeY : aEe, oEo = D == G::PERM_D ? aEoY : aEo;
            const int na = off + dj * ps, nb = off + (N - 1 - dj) * ps;
            double qa[NV], qb[NV], aa[nz(NA)], ab[nz(NA)], Fa[NV], Fb[NV];
            if constexpr (ST != 2) {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    qa[v] = EXA_SLD(na + v * VS);
                    qb[v] = EXA_SLD(nb + v * VS);
                }
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    aa[k] = EXA_SLD(AXO + na + k * VS);
                    ab[k] = EXA_SLD(AXO + nb + k * VS);
                }
            }
            if constexpr (ST == 1) {
#pragma unroll
                for (int v = 0; v < NV; v++) { ext.qa[v] = qa[v]; ext.qb[v] = qb[v]; }
#pragma unroll
                for (int k = 0; k < NA; k++) { ext.aa[k] = aa[k]; ext.ab[k] = ab[k]; }
                return;
            }
            if constexpr (ST == 2) {
#pragma unroll
                for (int v = 0; v < NV; v++) { qa[v] = ext.qa[v]; qb[v] = ext.qb[v]; }
#pragma unroll
                for (int k = 0; k < NA; k++) { aa[k] = ext.aa[k]; ab[k] = ext.ab[k]; }
            }
            const double sc = D == 0 ? idx0 : (D == 1 ? idx1 : idx2);
            // XT: positions of the lane's two nodes, time of its level slot (pencil (a, b) and level slot decoded from the table's offset)
            [[maybe_unused]] double xa[3], xb[3], tl = 0.0;
            if constexpr (XT) {
                const int ls = off / SL, r = off - ls * SL;
                const int a = D == 0 ? r / G::PY : r / G::PX, b = D == 0 ? r % G::PY : (D == 1 ? r % G::PX : (r % G::PX) / G::PY);
                const double fa = xi_of<N>(geo, a), fb = xi_of<N>(geo, b), ja = xi_of<N>(geo, dj), jb = xi_of<N>(geo, N - 1 - dj);
                xa[0] = xc[0] + (D == 0 ? ja : fa) * geo.h[0];
                xa[1] = xc[1] + (D == 1 ? ja : (D == 0 ? fa : fb)) * geo.h[1];
                xa[2] = xc[2] + (D == 2 ? ja : fb) * geo.h[2];
#pragma unroll
                for (int c = 0; c < 3; c++) xb[c] = xa[c];
                xb[D] = xc[D] + jb * geo.h[D];
                tl = ls ? tB : tA;
            }
            if constexpr (MODE == 0) {
                if constexpr (FXT) {
                    dg_flux_xt<PDE>(qa, xa, tl, D, Fa);
                    dg_flux_xt<PDE>(qb, xb, tl, D, Fb);
#pragma unroll
                    for (int v = 0; v < NV; v++) { Fa[v] *= sc; Fb[v] *= sc; }
                } else {
#ifdef EXA_M8_ABL_NOFLUX       // (timing ablations, never in the product: results are wrong)
#pragma unroll
                    for (int v = 0; v < NV; v++) { Fa[v] = qa[v] * sc + aa[v % nz(NA)]; Fb[v] = qb[v] * sc + ab[v % nz(NA)]; }
#else
                    PDE::template flux_scaled<D>(qa, aa, sc, Fa);
                    PDE::template flux_scaled<D>(qb, ab, sc, Fb);
#endif
                }
            }
            [[maybe_unused]] double na_[NV], nb_[NV];                   // B_D(q) (D q) / h_D at the lane's two nodes
            if constexpr (NCP) {
                double ga[NV], gb[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const double Pq = __builtin_amdgcn_mfma_f64_4x4x4f64(oEe, qa[v] + qb[v], 0.0, 0, 0, 0);
                    const double Mq = __builtin_amdgcn_mfma_f64_4x4x4f64(oEo, qa[v] - qb[v], 0.0, 0, 0, 0);
                    ga[v] = sc * (Mq + Pq);
                    gb[v] = sc * (Mq - Pq);
                    na_[v] = 0.0;
                    nb_[v] = 0.0;
                }
                dg_ncp<PDE>(qa, ga, xa, tl, D, na_);
                dg_ncp<PDE>(qb, gb, xb, tl, D, nb_);
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                double sa = 0.0, sb = 0.0;
                if constexpr (MODE == 0) {
                    const double e = Fa[v] + Fb[v], o = Fa[v] - Fb[v];
#ifdef EXA_M8_ABL_NOMFMA
                    const double Pv = oEe * e, Mv = oEo * o;
#else
                    const double Pv = __builtin_amdgcn_mfma_f64_4x4x4f64(oEe, e, 0.0, 0, 0, 0);
                    const double Mv = __builtin_amdgcn_mfma_f64_4x4x4f64(oEo, o, 0.0, 0, 0, 0);
#endif
                    sa = Mv + Pv;
                    sb = Mv - Pv;
                }
                if constexpr (NCP) {
                    sa += na_[v];
                    sb += nb_[v];
                }
                if constexpr (D == 0 && !EXA_M8_XINPLACE) {
                    hx[0][v] = sa;
                    hx[1][v] = sb;
                } else {
#ifdef EXA_M8_ABL_NOSTORE
                    asm volatile("" ::"v"(sa), "v"(sb));
#elif EXA_M8_LAYOUT
                    if constexpr (D == 2) {                            // z: the compact array, 16 consecutive doubles per store group
                        const int zb = 2 * QSZ + v * G::ZV + (wave >> 2) * G::ZL + (wave & 3) * 128 + (lane & 15);
                        lds[zb + dj * 16] = sa;
                        lds[zb + (N - 1 - dj) * 16] = sb;
                    } else {                                           // y: over Q (in place); x: array A
                        lds[(D == 0 ? QSZ : 0) + na + v * VS] = sa;
                        lds[(D == 0 ? QSZ : 0) + nb + v * VS] = sb;
                    }
#else
                    lds[D * QSZ + na + v * VS] = sa;
                    lds[D * QSZ + nb + v * VS] = sb;
#endif
                }
            }
        }
    };
    auto round = [&](auto dc, int tb, double (&hx)[2][NV], double tA, double tB, auto mode) {
        RoundOps none;
        round_staged(dc, tb, hx, tA, tB, mode, std::integral_constant<int, 0>{}, none);
    };
#ifndef EXA_M8_XEARLY        // 1: the in-place round's operands requested in front of the barrier (r5).  Measured: 89.6 against 88.3 ms per 64^3 launch (no more
#define EXA_M8_XEARLY 0      // registers, no spill -- the loads merely compete with the stores of the rounds in front of the barrier): off.
#endif
#ifndef EXA_M8_PIPE          // (measured SLOWER: 117.3 against 98.2 ms per 64^3 launch -- the 28 VGPRs of the next round's operands do not exist: 204 B of scratch)
#define EXA_M8_PIPE 0
#endif
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    auto derive = [&](int tb, double tA, double tB, auto mode) {      // tb: 0 = steps of two levels, 1 = iteration 0 (one level)
        double hx[2][NV];
#ifdef EXA_M8_SKEW            // (experiment: the second wave of every SIMD starts its rounds EXA_M8_SKEW x 64 cycles late, so that its LDS phases fall beside the other's arithmetic)
        if (wave >= 4) __builtin_amdgcn_s_sleep(EXA_M8_SKEW);
#endif
#if EXA_M8_PIPE
        static_assert(!XT && !NCP && !EXA_M8_LAYOUT, "EXA_M8_PIPE: built-in term sets, r3 layout only");
        // software-pipelined over the three rounds: the LDS loads of the next round are issued in front of the matrix instructions and the
        // stores of the current one (they read Q and the scalars, which no round writes), so their latency and the store queue overlap
        const int t_ = opaque_v(tid), dj = opaque_v(d_j);
        struct Ops { double qa[NV], qb[NV], aa[nz(NA)], ab[nz(NA)]; };
        auto offs = [&](auto dc, int& na, int& nb) -> bool {
            constexpr int D = decltype(dc)::value;
            const int off = ltab[(tb * 3 + D) * NT + t_];
            const bool have = off != 0xffff;                          // (wave-uniform: a wave has 16 pencils or none)
            const int o2 = have ? off : 0;
            na = o2 + dj * G::pstride(D);
            nb = o2 + (N - 1 - dj) * G::pstride(D);
            return have;
        };
        auto load = [&](int na, int nb, Ops& r) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                r.qa[v] = EXA_SLD(na + v * VS);
                r.qb[v] = EXA_SLD(nb + v * VS);
            }
#pragma unroll
            for (int k = 0; k < NA; k++) {
                r.aa[k] = EXA_SLD(AXO + na + k * VS);
                r.ab[k] = EXA_SLD(AXO + nb + k * VS);
            }
        };
        auto evenodd = [&](auto dc, const Ops& r, double (&e)[NV], double (&o)[NV]) {
            constexpr int D = decltype(dc)::value;
            double Fa[NV], Fb[NV];
            const double sc = D == 0 ? idx0 : (D == 1 ? idx1 : idx2);
            PDE::template flux_scaled<D>(r.qa, r.aa, sc, Fa);
            PDE::template flux_scaled<D>(r.qb, r.ab, sc, Fb);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                e[v] = Fa[v] + Fb[v];
                o[v] = Fa[v] - Fb[v];
            }
        };
        auto matrix = [&](auto dc, bool have, int na, int nb, const double (&e)[NV], const double (&o)[NV]) {
            constexpr int D = decltype(dc)::value;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const double Pv = __builtin_amdgcn_mfma_f64_4x4x4f64(aEe, e[v], 0.0, 0, 0, 0);
                const double Mv = __builtin_amdgcn_mfma_f64_4x4x4f64(aEo, o[v], 0.0, 0, 0, 0);
                if constexpr (D == 0) {
                    hx[0][v] = Mv + Pv;
                    hx[1][v] = Mv - Pv;
                } else if (have) {
                    lds[D * QSZ + na + v * VS] = Mv + Pv;
                    lds[D * QSZ + nb + v * VS] = Mv - Pv;
                }
            }
        };
        using D0 = std::integral_constant<int, 0>;
        using D1 = std::integral_constant<int, 1>;
        using D2 = std::integral_constant<int, 2>;
        int na1, nb1, na2, nb2, na0, nb0;
        const bool h1 = offs(D1{}, na1, nb1), h2 = offs(D2{}, na2, nb2), h0 = offs(D0{}, na0, nb0);
        (void)h0;
        Ops r;
        double e[NV], o[NV];
        load(na1, nb1, r);
        evenodd(D1{}, r, e, o);
        load(na2, nb2, r);                                            // (round z's operands: in flight under round y's matrix instructions and stores)
        matrix(D1{}, h1, na1, nb1, e, o);
        evenodd(D2{}, r, e, o);
        load(na0, nb0, r);
        matrix(D2{}, h2, na2, nb2, e, o);
        evenodd(D0{}, r, e, o);
        matrix(D0{}, true, 0, 0, e, o);                               // x last: its sums wait in registers for the barrier
#elif EXA_M8_XINPLACE
        // r5: the x round runs BEHIND the barrier that ends the y and z rounds and stores its sums straight over Q -- the four lanes of an x pencil (one wave) are
        // the only readers of the nodes they overwrite, and they have read them (in-order LDS) before their matrix instructions complete.  The separate
        // "Q := S_x" phase (10 stores per lane with nothing beside them, 955 of the 7 800 cycles of a step in the stamp build) and the ten values held over the
        // barrier are gone; the barrier count is the same.
#if EXA_M8_PLANE
        // plane ownership: z on the wave's own nodes (no barrier behind the puts, which were this wave's own: LDS operations of a wave complete in order),
        // y between the two barriers of the step, x in place on the wave's own nodes, and the fold straight behind it
        EXA_M8_WAVE_ORDER();
        round(std::integral_constant<int, 2>{}, tb, hx, tA, tB, mode);
        EXA_RSTAMP(0);
        __syncthreads();                                              // every put of the step is done
        EXA_RSTAMP(1);
        round(std::integral_constant<int, 1>{}, tb, hx, tA, tB, mode);
        EXA_RSTAMP(2);
        __syncthreads();                                              // every y round has read Q; S_y (A) is complete
        EXA_RSTAMP(3);
        round(std::integral_constant<int, 0>{}, tb, hx, tA, tB, mode);
        EXA_M8_WAVE_ORDER();
        return;
#endif
        // (EXA_M8_LAYOUT: the direction in place is y, the one whose stores are conflict-free in the layout of Q; before r5's layout it was x)
        constexpr int DL = EXA_M8_LAYOUT ? 1 : 0, DF = EXA_M8_LAYOUT ? 0 : 1;
        round(std::integral_constant<int, DF>{}, tb, hx, tA, tB, mode);
        EXA_RSTAMP(0);
        round(std::integral_constant<int, 2>{}, tb, hx, tA, tB, mode);
        EXA_RSTAMP(1);
#if EXA_M8_XEARLY
        RoundOps xo;
        round_staged(std::integral_constant<int, DL>{}, tb, hx, tA, tB, mode, std::integral_constant<int, 1>{}, xo);
        __syncthreads();                                              // every read of Q by the first two rounds is done (this round's own are in its registers)
        EXA_RSTAMP(2);
        round_staged(std::integral_constant<int, DL>{}, tb, hx, tA, tB, mode, std::integral_constant<int, 2>{}, xo);
#else
        __syncthreads();                                              // every read of Q by the first two rounds is done
        EXA_RSTAMP(2);
        round(std::integral_constant<int, DL>{}, tb, hx, tA, tB, mode);
#endif
        EXA_RSTAMP(3);
        __syncthreads();                                              // the three directional sums are complete
        return;
#else
#ifdef EXA_M8_ROUND_STAMPS     // (stamp builds: the rounds one by one in slots 0..3 -- the iteration-0 slots then hold rounds as well)
        round(std::integral_constant<int, 1>{}, tb, hx, tA, tB, mode);
        EXA_STAMP(0);
        round(std::integral_constant<int, 2>{}, tb, hx, tA, tB, mode);
        EXA_STAMP(1);
        round(std::integral_constant<int, 0>{}, tb, hx, tA, tB, mode);
        EXA_STAMP(2);
        __syncthreads();
        EXA_STAMP(3);
#else
        round(std::integral_constant<int, 1>{}, tb, hx, tA, tB, mode);
        round(std::integral_constant<int, 2>{}, tb, hx, tA, tB, mode);
        round(std::integral_constant<int, 0>{}, tb, hx, tA, tB, mode);      // x last: its sums wait in registers for the barrier
#endif
#endif
#ifndef EXA_M8_ROUND_STAMPS
        __syncthreads();                                              // every read of Q is done; S_y (A), S_z (B) are complete
#endif
        const int off = ltab[(tb * 3 + 0) * NT + opaque_v(tid)];
        if (off != 0xffff) {                                          // Q := S_x
            constexpr int ps = G::pstride(0);
            const int dj = opaque_v(d_j);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                lds[off + dj * ps + v * VS] = hx[0][v];
                lds[off + (N - 1 - dj) * ps + v * VS] = hx[1][v];
            }
        }
        __syncthreads();
    };
    auto put_level = [&](int ls, const double (&qv)[NV]) {
        double a[nz(NA)];
        PDE::aux_fast(qv, a);
#pragma unroll
        for (int v = 0; v < NV; v++) lds[o_off + ls * SL + v * VS] = qv[v];
#pragma unroll
        for (int k = 0; k < NA; k++) lds[AXO + o_off + ls * SL + k * VS] = a[k];
    };

#ifndef EXA_M8_PREFETCH_U    // 1: the next cell's u requested behind the time averages of this one (in flight under the volume / trace / u* phases).
#define EXA_M8_PREFETCH_U 0  // Measured r5: 89.5 against 88.4 ms per 64^3 launch -- the ten registers it holds cost 4 spilled VGPRs; off.
#endif
    [[maybe_unused]] double un[NV];
    if constexpr (EXA_M8_PREFETCH_U) {
        if ((long)blockIdx.x < box.nbox) {
            const double* p0 = u_in + (box.cell(blockIdx.x) * NN + o_n) * NV;
#pragma unroll
            for (int v = 0; v < NV; v++) un[v] = p0[v];
        }
    }
    for (long b = blockIdx.x; b < box.nbox; b += gridDim.x) {
        const long cell = box.cell(b);
        if constexpr (XT) {
            const long cz = b % box.nb[2], cy = (b / box.nb[2]) % box.nb[1], cx = b / (box.nb[2] * box.nb[1]);
            xc[0] = geo.x0[0] + (double)(box.lo[0] + cx) * geo.h[0];
            xc[1] = geo.x0[1] + (double)(box.lo[1] + cy) * geo.h[1];
            xc[2] = geo.x0[2] + (double)(box.lo[2] + cz) * geo.h[2];
        }
        const double* up = u_in + (cell * NN + o_n) * NV;             // (u is re-read where needed: an L2 hit against 10 VGPRs of a kernel at its cap)
        double u[NV], q[N][NV];
#pragma unroll
        for (int v = 0; v < NV; v++) u[v] = EXA_M8_PREFETCH_U ? un[v] : up[v];

        // (XT: the terms depend on the level time -- every iteration is a full one, started from q_l = u)
        if constexpr (XT) {
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = u[v];
        } else {
            // ---- Picard iteration 0: the iterate is constant in time -- one level, row sums of T
            put_level(0, u);
            EXA_STAMP(0);
            EXA_M8_SYNC_PUTS();
            EXA_STAMP(1);
            {
                derive(1, 0.0, 0.0, M0{});                               // iteration 0 has its own table (one level: four waves per round)
            }
            EXA_STAMP(2);
            {
                double S[NV], Ts[N];
                sload<N>(step_here<N>(step_raw)->Tsdt, Ts);
    #pragma unroll
                for (int v = 0; v < NV; v++) S[v] = sums_at(0, v);
                if constexpr (pde_has_source<PDE>::value) {
                    double Sq[NV];
                    source_at(u, 0, Sq);
    #pragma unroll
                    for (int v = 0; v < NV; v++) S[v] -= Sq[v];
                }
    #pragma unroll
                for (int l = 0; l < N; l++)
    #pragma unroll
                    for (int v = 0; v < NV; v++) q[l][v] = fma(Ts[l], S[v], u[v]);
            }
            EXA_STAMP(3);

        }
        // ---- Picard iterations 1 .. n_it - 1: steps of two levels; the load of the next step's levels sits inside the fold
        auto load_levels = [&](auto lc, const double (&qq)[N][NV]) {
            constexpr int l0 = decltype(lc)::value;
#pragma unroll
            for (int ls = 0; ls < 2; ls++) put_level(ls, qq[l0 + ls]);
        };
        constexpr int IT0 = XT ? 0 : 1;                                // first full iteration
        if (n_it > IT0) load_levels(std::integral_constant<int, 0>{}, q);
#ifndef EXA_M8_DEFER
#define EXA_M8_DEFER 1
#endif
#if EXA_M8_DEFER
        // r5: the time contraction of an iteration is DEFERRED to its end.  A step only sums S_x + S_y + S_z of its two levels into registers; the
        // FMAs acc[l'] = u - dt sum_l T[l'][l] S_l run once all levels are there (same order in l as before: bit-identical).  Live through the derive
        // rounds of step st: the levels of q not yet in LDS (10 (3 - st) doubles) + the sums so far (10 st) = 30 doubles instead of up to 60 (iterate
        // + all eight accumulators), which is what the register allocator spilled around (10 VGPRs, 44 B of scratch, one reload per iteration).
        for (int it = IT0; it < n_it; it++) {
            double Sk[N][NV];
            [[maybe_unused]] double uu[NV];
            static_for<0, LS>([&](auto sc_) {
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = 2 * st;
                EXA_STAMP(4);
                EXA_M8_SYNC_PUTS();
                EXA_STAMP(5);
                derive(0, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M0{});
                EXA_STAMP(6);
                if constexpr (st + 1 == LS) {                          // u for the contraction below: requested in front of the sums' loads
#pragma unroll
                    for (int v = 0; v < NV; v++) uu[v] = up[v];
                }
#pragma unroll
                for (int ls = 0; ls < 2; ls++)
#pragma unroll
                    for (int v = 0; v < NV; v++) Sk[l0 + ls][v] = sums_at(ls, v);
                if constexpr (pde_has_source<PDE>::value) {           // q_t + div F = S(q): on the iterate in the owner's registers
#pragma unroll
                    for (int ls = 0; ls < 2; ls++) {
                        double Sq[NV];
                        source_at(q[l0 + ls], l0 + ls, Sq);
#pragma unroll
                        for (int v = 0; v < NV; v++) Sk[l0 + ls][v] -= Sq[v];
                    }
                }
                if constexpr (st + 1 < LS) load_levels(std::integral_constant<int, l0 + 2>{}, q);
                EXA_STAMP(7);
            });
            static_for<0, LS>([&](auto sc_) {                          // q := u - dt T S, two input levels per batch of scalar operands
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = 2 * st;
                double Tm[2 * N];                                      // -dt T[l'][l0 + ls], l' fastest
                sload<2 * N>(step_here<N>(step_raw)->TdtT + l0 * N, Tm);
#pragma unroll
                for (int ls = 0; ls < 2; ls++)
#pragma unroll
                    for (int lp = 0; lp < N; lp++)
#pragma unroll
                        for (int v = 0; v < NV; v++) q[lp][v] = fma(Tm[ls * N + lp], Sk[l0 + ls][v], (st == 0 && ls == 0) ? uu[v] : q[lp][v]);
            });
            if (it + 1 < n_it) load_levels(std::integral_constant<int, 0>{}, q);
            EXA_STAMP(4);
        }
#else
        for (int it = IT0; it < n_it; it++) {
            double acc[N][NV];
            static_for<0, LS>([&](auto sc_) {
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = 2 * st;
                EXA_STAMP(4);
                EXA_M8_SYNC_PUTS();
                EXA_STAMP(5);
#ifdef EXA_M8_EARLY_U
                [[maybe_unused]] double uu[NV];
                if constexpr (st == 0) {
#pragma unroll
                    for (int v = 0; v < NV; v++) uu[v] = up[v];
                }
#endif
                derive(0, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M0{});
                EXA_STAMP(6);
                double Tm[2 * N];                                      // -dt T[l'][l0 + ls], l' fastest
                sload<2 * N>(step_here<N>(step_raw)->TdtT + l0 * N, Tm);
#ifndef EXA_M8_EARLY_U
                [[maybe_unused]] double uu[NV];
                if constexpr (st == 0) {
#pragma unroll
                    for (int v = 0; v < NV; v++) uu[v] = up[v];
                }
#endif
                double Sx[2][NV];                                      // (summed as they arrive: 160 VGPRs of iterate + accumulators leave no room for 30 loads in flight)
#pragma unroll
                for (int ls = 0; ls < 2; ls++)
#pragma unroll
                    for (int v = 0; v < NV; v++) Sx[ls][v] = sums_at(ls, v);
                if constexpr (pde_has_source<PDE>::value) {           // q_t + div F = S(q): on the iterate in the owner's registers
#pragma unroll
                    for (int ls = 0; ls < 2; ls++) {
                        double Sq[NV];
                        source_at(q[l0 + ls], l0 + ls, Sq);
#pragma unroll
                        for (int v = 0; v < NV; v++) Sx[ls][v] -= Sq[v];
                    }
                }
                if constexpr (st + 1 < LS) load_levels(std::integral_constant<int, l0 + 2>{}, q);
#pragma unroll
                for (int ls = 0; ls < 2; ls++)
#pragma unroll
                    for (int lp = 0; lp < N; lp++)
#pragma unroll
                        for (int v = 0; v < NV; v++) {
                            if constexpr (st == 0) acc[lp][v] = fma(Tm[ls * N + lp], Sx[ls][v], ls == 0 ? uu[v] : acc[lp][v]);
                            else acc[lp][v] = fma(Tm[ls * N + lp], Sx[ls][v], acc[lp][v]);
                        }
                if constexpr (st + 1 == LS) {
                    if (it + 1 < n_it) load_levels(std::integral_constant<int, 0>{}, acc);
                }
                EXA_STAMP(7);
            });
#pragma unroll
            for (int l = 0; l < N; l++)
#pragma unroll
                for (int v = 0; v < NV; v++) q[l][v] = acc[l][v];
        }

#endif

        // ---- NCP: the time-averaged non-conservative term of the FINAL iterate enters u* point-wise: one more pass of the derive rounds over the
        // levels, non-conservative part only, weighted with w_l by the owners
        [[maybe_unused]] double pw[NV];
        if constexpr (NCP) {
#pragma unroll
            for (int v = 0; v < NV; v++) pw[v] = 0.0;
            load_levels(std::integral_constant<int, 0>{}, q);
            static_for<0, LS>([&](auto sc_) {
                constexpr int st = decltype(sc_)::value;
                constexpr int l0 = 2 * st;
                EXA_M8_SYNC_PUTS();
                derive(0, XT ? level_t(l0) : 0.0, XT ? level_t(l0 + 1) : 0.0, M1{});
                double wl[2];
                sload<2>(ops_here<N>(ops_raw)->w + l0, wl);
#pragma unroll
                for (int ls = 0; ls < 2; ls++)
#pragma unroll
                    for (int v = 0; v < NV; v++) pw[v] = fma(wl[ls], sums_at(ls, v), pw[v]);
                if constexpr (st + 1 < LS) load_levels(std::integral_constant<int, l0 + 2>{}, q);
            });
        }
        // ---- time averages: qbar | Fbar_x | Fbar_y | Fbar_z (| time-averaged source)
        __syncthreads();                                              // every fold has read its sums: the closing image reuses the LDS
        {
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
                    double S1[NV];
                    source_at(q[l], l, S1);
#pragma unroll
                    for (int v = 0; v < NV; v++) Sbar[v] += wm[l] * S1[v];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                lds[v * SL + o_off] = qb[v];
#pragma unroll
                for (int d = 0; d < DIM; d++) lds[((1 + d) * NV + v) * SL + o_off] = Fb[d][v];
                if constexpr (pde_has_source<PDE>::value) lds[(4 * NV + v) * SL + o_off] = Sbar[v];
            }
        }
        EXA_STAMP(8);
        if constexpr (EXA_M8_PREFETCH_U) {
            if (b + gridDim.x < box.nbox) {
                const double* pn = u_in + (box.cell(b + gridDim.x) * NN + opaque_v(o_n)) * NV;
#pragma unroll
                for (int v = 0; v < NV; v++) un[v] = pn[v];
            }
        }
        __syncthreads();

        // ---- volume integral (in place over Fbar_d) + face extrapolation: pencil tasks (d, v, t), t fastest
        {
            // (the operator entries are fetched next to their use: KEO | 1/w | phiL | phiR together are 61 doubles, more than the SGPR file)
            const EXA_AS4 DgOps<N>* o = ops_here<N>(ops_raw);
            for (int task = tid; task < DIM * NV * NF; task += NT) {
                const int d = task / (NV * NF);
                const int r = task - d * (NV * NF);
                const int v = r / NF, t = r - v * NF;
                const int ps = G::pstride(d);
                const int pb = G::pbase(d, t);
                double qb[N], Fb[N], vol[N];
#pragma unroll
                for (int j = 0; j < N; j++) {
                    qb[j] = EXA_SLD(v * SL + pb + j * ps);
                    Fb[j] = EXA_SLD(((1 + d) * NV + v) * SL + pb + j * ps);
                }
                eo_apply<N>(o->KEO, Fb, vol);
                const double sc = dt * (d == 0 ? idx0 : (d == 1 ? idx1 : idx2));
#pragma unroll
                for (int i = 0; i < N; i++) lds[((1 + d) * NV + v) * SL + pb + i * ps] = sc * o->iw[i] * vol[i];
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
        EXA_STAMP(9);
        __syncthreads();

        // ---- u* = u + sum_d vol_d (+ dt * time-averaged source): each owner its node
        {
            double* uo = u_out + (cell * NN + o_n) * NV;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                double us = up[v];
                if constexpr (NCP) us -= dt * pw[v];
                if constexpr (pde_has_source<PDE>::value) us += dt * EXA_SLD((4 * NV + v) * SL + o_off);
#pragma unroll
                for (int d = 0; d < DIM; d++) us += EXA_SLD(((1 + d) * NV + v) * SL + o_off);
                uo[v] = us;
            }
        }
        __syncthreads();                                              // LDS is reused by the next cell
        EXA_STAMP(10);
    }
    EXA_STAMP_FLUSH();
}

}  // namespace exa
