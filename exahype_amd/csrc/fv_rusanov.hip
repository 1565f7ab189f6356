// Fused Finite-Volume Rusanov patch update for gfx950 -- the device form of the
// reference's generated kernel `time_step` (`Unit test/test.cpp:3-111`, what
// `exahype/printers/CPPPrinter.py:84-90` emits for the statement list of
// `examples/Batched_stateless.py:25-35`).
//
// The reference runs ten separate sweeps over heap temporaries (Q_copy,
// tmp_flux_{x,y}, tmp_eigen_{x,y}); here one workgroup owns one patch, every
// thread evaluates the whole stencil of its volumes in registers (flux and
// eigenvalue of the 2*dim neighbours are recomputed instead of staged), a
// workgroup barrier separates the last read of the patch from the first write,
// and the result goes back in place -- Q is read once and written once.
//
// MODE 0 (faithful): exactly the arithmetic and evaluation order of
//   test.cpp:60-95 (no dt/h, dissipation on variable 0 only, reference sign),
//   with the temporaries the reference leaves uninitialised taken as zero
//   (SURVEY.md F6).  This unit is compiled with -ffp-contract=off so the result
//   is bit-identical to the g++ build of the reference wherever that is defined.
// MODE 1 (corrected Rusanov): dt/h, all n_real variables.
#include <cstdio>
#include <type_traits>
#include "exa_launch.hpp"
#include "exa_pde.hpp"
#ifdef EXA_USER_PDE_HEADER
#include EXA_USER_PDE_HEADER      // struct exa::UserPDE (exahype_amd/pde_codegen.py)
#endif

namespace exa {

constexpr int MAXV = 8;
typedef double v2d __attribute__((ext_vector_type(2)));

template <int I, int E, class F> __device__ inline void static_for_fv(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for_fv<I + 1, E>(f);
    }
}

// What the `exahype2::CellData` flavour of the kernel adds to (Q, dt) (`exahype/KernelBuilder.py:217-218`, `Unit test/correctness_test.cpp:142`):
// a separate, halo-less output array, the patch centres and the time.  All null / zero for the in-place call.
struct FvCellData {
    double* out;              // QOut[patch][P^dim][n_real + n_aux], or null: Q is updated in place
    const double* centre;     // [patch][dim] cell centres, or null: every patch is centred at the origin
    double t, h;              // time; volume size
    // GRID step (exa_fv_grid_step_device; the enclave task's halo fill folded into the patch update): the patches are the cells of a Cartesian
    // grid g[0] x g[1] (x g[2]), patch index row-major, and BOTH arrays are halo-less -- Q [patch][P^dim][V] is only read, `out` (same layout)
    // receives the new states.  What the stencil needs beyond a patch face comes from the adjacent interior layers of the face neighbour
    // (periodic wrap) or, on a domain face of a non-periodic grid, is the prescribed state bstate[(axis * 2 + side) * V ..].
    const double* bstate;     // null: periodic
    int g[3];
    double* lam;              // grid step, optional: lam[0] = max(lam[0], largest eigenvalue of the NEW interior states over the directions) -- the
                              // CFL scan of the next step without a pass of its own (integer atomic max: zeroed by the launcher)
    int grid_on;
};
// logical block of workgroup b when each of the 8 XCDs (which take the workgroups of a launch round-robin) is to work on a contiguous range
__device__ inline long fv_xcd_contiguous(long b, long n) {
    const long per = n / 8;
    return b < per * 8 ? (b % 8) * per + b / 8 : b;
}
// wave-level maximum -> one atomic per wave (non-negative doubles order like their bit patterns)
__device__ inline void fv_lam_commit(double* lam, double mx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = nan_max(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(lam), (unsigned long long)__double_as_longlong(mx));
}

// grid step: where the stencil finds the state of volume `co` (coordinates with halo, exactly one of them in a halo layer: co[a] < H or >= P + H)
// of patch `patch` -- the interior volume of the face neighbour it mirrors, or the boundary state of that domain face
// grid coordinates of a patch (patch index row-major; 32-bit arithmetic: a 64-bit division costs ~100 instructions)
template <int DIM>
__device__ inline void fv_grid_coords(const FvCellData& cd, long patch, int* pg) {
    unsigned r = (unsigned)patch;
    if constexpr (DIM == 3) { pg[2] = (int)(r % (unsigned)cd.g[2]); r /= (unsigned)cd.g[2]; } else pg[2] = 0;
    pg[1] = (int)(r % (unsigned)cd.g[1]);
    pg[0] = (int)(r / (unsigned)cd.g[1]);
}
template <int DIM>
__device__ inline bool fv_grid_locate(const FvCellData& cd, const int* pgc, const int* co, int a, int P, int H, long& np, int* cn) {
    int pg[3] = {pgc[0], pgc[1], pgc[2]};
    const int side = co[a] < H ? 0 : 1;
    if (cd.bstate && (side == 0 ? pg[a] == 0 : pg[a] == cd.g[a] - 1)) { np = a * 2 + side; return true; }    // domain face: the prescribed state
    pg[a] += side == 0 ? -1 : 1;
    if (pg[a] < 0) pg[a] += cd.g[a];
    if (pg[a] >= cd.g[a]) pg[a] -= cd.g[a];
    np = DIM == 3 ? ((long)pg[0] * cd.g[1] + pg[1]) * cd.g[2] + pg[2] : (long)pg[0] * cd.g[1] + pg[1];
#pragma unroll
    for (int b = 0; b < DIM; b++) cn[b] = b == a ? (side == 0 ? co[b] + P : co[b] - P) : co[b];     // the interior volume it mirrors
    return false;
}
// ... as an address in the halo-less array Q [patch][P^DIM][V]
template <int DIM>
__device__ inline const double* fv_grid_source(const double* Q, const FvCellData& cd, const int* pgc, const int* co, int a, int P, int H, int V) {
    long np;
    int cn[3];
    if (fv_grid_locate<DIM>(cd, pgc, co, a, P, H, np, cn)) return cd.bstate + np * V;
    long c = 0;
#pragma unroll
    for (int b = 0; b < DIM; b++) c = c * P + (cn[b] - H);
    return Q + (np * (DIM == 3 ? (long)P * P * P : (long)P * P) + c) * V;
}
// the e-th face-halo volume of a patch (e < 2 * DIM * H * P^(DIM-1); corners and edges are not part of the 2 DIM + 1-point stencil):
// coordinates with halo and the axis it lies beyond
template <int DIM>
__device__ inline void fv_halo_volume(int e, int P, int H, int* co, int& a) {
    const int per = DIM == 3 ? H * P * P : H * P;             // volumes of one face's halo slab
    const int f = e / per;
    int r = e - f * per;
    a = f >> 1;
    const int layer = r % H;
    r /= H;
    int t[2] = {r % P, r / P};                                 // transverse interior coordinates
    int k = 0;
#pragma unroll
    for (int b = 0; b < DIM; b++) co[b] = b == a ? ((f & 1) ? P + H + layer : layer) : t[k++] + H;
}

// STAGE = true (small patches): the workgroup's `ppb` patches are one contiguous block of HBM; it is
// copied into LDS with coalesced 16-byte loads and the stencil reads LDS (the AoS stencil reads straight
// from HBM touch ~40 cache lines per wave instruction: 1.7 TB/s; staged: see DESIGN.md 4.3).
// SHAPE: compile-time (patch_size, halo_size, n_real, n_real + n_aux), or all zero for run-time values.  The reference's
// own configuration (4, 1, 5, 10) is built specialised: the variable masks and the 64-bit index arithmetic fold away
// (a wave executed ~1 030 VALU instructions in the generic build; the arithmetic and its order are the same).
#ifndef EXA_FV_STORE
#define EXA_FV_STORE 1
#endif
__device__ inline void fv_row_store(v2d val, v2d* at) {
#if EXA_FV_STORE == 0
    *at = val;
#elif EXA_FV_STORE == 1
    __builtin_nontemporal_store(val, at);
#elif EXA_FV_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(at), "v"(val) : "memory");
#elif EXA_FV_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(at), "v"(val) : "memory");
#elif EXA_FV_STORE == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(at), "v"(val) : "memory");
#elif EXA_FV_STORE == 5
    asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(at), "v"(val) : "memory");
#endif
}
template <int TP, int TH, int TM, int TV> struct FvShape {
    static constexpr int P = TP, H = TH, M = TM, V = TV;
};
using FvRuntimeShape = FvShape<0, 0, 0, 0>;

// PERSIST (staged small patches only): the grid is as many workgroups as the chip holds; a workgroup walks over the blocks
// of `ppb` patches and requests the NEXT block into registers (16-byte loads at clamped indices, no load under a lane
// condition) before it updates the current one from LDS -- the 46 KB read of a block overlaps the stencil and the
// write-back of its predecessor instead of standing alone in front of a barrier.
constexpr int FV_HR = 16;          // double2 per thread that hold a block in flight (<= 64 KiB per block at 256 threads)
#ifndef EXA_FV_GRID_WAVES
#define EXA_FV_GRID_WAVES 1          // persistent grid step: no register cap (199 VGPRs, two workgroups per CU: 0.855 ms per 2^20-patch step); capped at 168 for
#endif                               // three workgroups per CU it spills 31 registers and takes 0.995 ms
template <int DIM, class PDE, int MODE, int CPT, int NT, bool STAGE, class SHAPE = FvRuntimeShape, bool PERSIST = false, bool GRID = false>
__global__ void __launch_bounds__(NT, ((GRID && PERSIST && NT == 256) ? EXA_FV_GRID_WAVES : 1))
fv_rusanov_kernel(double* __restrict__ Q, int P_rt, int H_rt, int m_rt, int V_rt, double dt, double dt_over_h, long n_patches,
                  int ppb, const long* __restrict__ slot, FvCellData cd) {
    const int P = SHAPE::P ? SHAPE::P : P_rt, H = SHAPE::P ? SHAPE::H : H_rt;
    const int m = SHAPE::P ? SHAPE::M : m_rt, V = SHAPE::P ? SHAPE::V : V_rt;
    extern __shared__ __attribute__((aligned(16))) double fv_lds[];
    // `ppb` small patches share one workgroup (the reference's 4x4 patch has 16 volumes: one patch per
    // 256-thread workgroup would idle 94 % of the lanes); large patches use ppb = 1 and CPT volumes per thread.
    const int S = P + 2 * H;
    const long vol = (DIM == 3) ? (long)S * S * S : (long)S * S;
    const int ncell = (DIM == 3) ? P * P * P : P * P;
    const int pl = (CPT == 1) ? (int)threadIdx.x / ncell : 0;            // patch slot of this thread
    static_assert(!PERSIST || (STAGE && CPT == 1), "the persistent form is the staged small-patch kernel");
    const long nblk = (n_patches + ppb - 1) / ppb;
    const long blk_step = PERSIST ? (long)gridDim.x : nblk;      // not persistent: exactly one pass, block = blockIdx.x
    // registers that hold a block in flight: exact for a compile-time shape (the reference's: 12), FV_HR otherwise
    constexpr int SPOW = SHAPE::P ? (DIM == 3 ? (SHAPE::P + 2 * SHAPE::H) * (SHAPE::P + 2 * SHAPE::H) * (SHAPE::P + 2 * SHAPE::H) : (SHAPE::P + 2 * SHAPE::H) * (SHAPE::P + 2 * SHAPE::H)) : 0;
    constexpr int PPOW = SHAPE::P ? (DIM == 3 ? SHAPE::P * SHAPE::P * SHAPE::P : SHAPE::P * SHAPE::P) : 1;
    constexpr int HRS = !PERSIST ? 1 : (SHAPE::P ? ((NT / PPOW) * (GRID ? PPOW : SPOW) * SHAPE::V / 2 + NT - 1) / NT : FV_HR);
    v2d hold[HRS];
    const long gvol = GRID ? (long)ncell : vol;                   // volumes per patch in the array Q (grid step: halo-less)
    auto request = [&](long blk) {                                // block blk -> registers (PERSIST: Q 16-byte aligned, even block size)
        const long first = blk * ppb;
        const long npatch = (n_patches - first < ppb) ? n_patches - first : ppb;
        const int npair = (int)(npatch * gvol * V / 2);
        const v2d* s2 = reinterpret_cast<const v2d*>(Q + first * gvol * V);
#pragma unroll
        for (int r = 0; r < HRS; r++) {
            const int xx = (int)threadIdx.x + r * NT;
#ifdef EXA_FV_NTLOAD
            hold[r] = __builtin_nontemporal_load(s2 + (xx < npair ? xx : npair - 1));
#else
            hold[r] = s2[xx < npair ? xx : npair - 1];
#endif
        }
    };
    // grid step: the face-halo states of a block.  The patches of a block are consecutive along the LAST grid axis, so the halos beyond the
    // faces of that axis are interior states of patches in the same LDS copy (an LDS -> LDS copy, `gather_local`) -- except for the low face of
    // the block's first patch, the high face of its last one and periodic wraps inside a block.  The halos beyond the faces of the other axes
    // (+ those two end faces) come from other blocks: they travel through registers like the block itself, all requests of a thread in flight
    // at once (a load-store loop serialises one memory latency per element).
    // grid coordinates of the block's patches: a table behind the LDS copy, [half][patch slot][3]; half = parity of the block's trip (the next
    // block's coordinates are written while the current block's are in use)
    [[maybe_unused]] int* pgtab = reinterpret_cast<int*>(fv_lds + (long)ppb * vol * V);
    [[maybe_unused]] auto fill_pgtab = [&](long blk, int half) {
        const long first = blk * ppb;
        if ((int)threadIdx.x < ppb && first + (int)threadIdx.x < n_patches) fv_grid_coords<DIM>(cd, first + (int)threadIdx.x, pgtab + (half * ppb + (int)threadIdx.x) * 3);
    };
    // double e of a halo-less block [patch slot][P^DIM][V] -> its place in the LDS copy (layout WITH halo, which the stencil reads)
    [[maybe_unused]] auto dense_to_lds = [&](int e) -> int {
        const int vi = e / V, v = e - vi * V;                        // volume index in the block, variable
        const int pp = vi / ncell, dv = vi - pp * ncell;
        int c;
        if constexpr (DIM == 3) c = ((dv / (P * P) + H) * S + (dv / P) % P + H) * S + dv % P + H;
        else c = (dv / P + H) * S + dv % P + H;
        return (int)(((long)pp * vol + c) * V) + v;
    };
    constexpr int PERC = SHAPE::P ? SHAPE::H * (DIM == 3 ? SHAPE::P * SHAPE::P : SHAPE::P) : 0;             // volumes of one face's halo slab
    constexpr int HGR = !GRID ? 1 : (SHAPE::P ? (((NT / PPOW) * (DIM - 1) + 1) * 2 * PERC * SHAPE::V + NT - 1) / NT : 12);
    // register slots of the decoded (full-block) path: units of GWC doubles; the generic path uses the same registers double by double
    constexpr int GWC = (SHAPE::P && SHAPE::V % 2 == 0) ? 2 : 1;
    constexpr int HG2 = (HGR + GWC - 1) / GWC;
    [[maybe_unused]] double hg[GWC * HG2];
    [[maybe_unused]] double lmax = 0.0;
    const int per = H * (DIM == 3 ? P * P : P);
    const int nh = 2 * DIM * per;
    const int nA_pp = (DIM - 1) * 2 * per * V;                       // remote halo states per patch (faces of the axes 0 .. DIM - 2)
    // halo volume hv (numbering of fv_halo_volume: faces axis-major) of patch slot pp -> place in the LDS copy, source
    [[maybe_unused]] auto halo_place = [&](int pp, int hv, int (&co)[3], int& ax) -> int {
        fv_halo_volume<DIM>(hv, P, H, co, ax);
        long c = 0;
#pragma unroll
        for (int b = 0; b < DIM; b++) c = c * S + co[b];
        return (int)(((long)pp * vol + c) * V);
    };
    // remote element e of a block: (patch slot, halo volume, variable)
    [[maybe_unused]] auto remote_elem = [&](int npatch, int e, int& pp, int& hv, int& v) {
        const int nmain = npatch * nA_pp;
        if (e < nmain) {
            pp = e / nA_pp;
            const int r = e - pp * nA_pp;
            hv = r / V;
            v = r - hv * V;
        } else {                                                     // low face of the first patch, high face of the last one (last axis)
            const int r = e - nmain, which = r / (per * V), r2 = r - which * (per * V);
            pp = which == 0 ? 0 : npatch - 1;
            hv = (2 * (DIM - 1) + which) * per + r2 / V;
            v = r2 % V;
        }
    };
    [[maybe_unused]] auto gather_request_any = [&](long blk, int half) {
        const long first = blk * ppb;
        const int npatch = (int)((n_patches - first < ppb) ? n_patches - first : ppb);
        const int ne = npatch * nA_pp + 2 * per * V;
#pragma unroll
        for (int r = 0; r < HGR; r++) {
            const int e0 = (int)threadIdx.x + r * NT, e = e0 < ne ? e0 : ne - 1;      // (clamped, unconditional: see request)
            int pp, hv, v, co[3] = {0, 0, 0}, ax;
            remote_elem(npatch, e, pp, hv, v);
            halo_place(pp, hv, co, ax);
            hg[r] = fv_grid_source<DIM>(Q, cd, pgtab + (half * ppb + pp) * 3, co, ax, P, H, V)[v];
        }
    };
    [[maybe_unused]] auto gather_land_any = [&](long blk, int half) {
        const long first = blk * ppb;
        const int npatch = (int)((n_patches - first < ppb) ? n_patches - first : ppb);
        const int ne = npatch * nA_pp + 2 * per * V;
#pragma unroll
        for (int r = 0; r < HGR; r++) {
            const int e = (int)threadIdx.x + r * NT;
            if (e < ne) {
                int pp, hv, v, co[3] = {0, 0, 0}, ax;
                remote_elem(npatch, e, pp, hv, v);
                fv_lds[halo_place(pp, hv, co, ax) + v] = hg[r];
            }
        }
        for (int e = (int)threadIdx.x + HGR * NT; e < ne; e += NT) {   // (run-time shapes with more remote halo states than the registers hold)
            int pp, hv, v, co[3] = {0, 0, 0}, ax;
            remote_elem(npatch, e, pp, hv, v);
            const int off = halo_place(pp, hv, co, ax);
            fv_lds[off + v] = fv_grid_source<DIM>(Q, cd, pgtab + (half * ppb + pp) * 3, co, ax, P, H, V)[v];
        }
    };
    // FULL blocks (ppb patches): which remote states a lane fetches, where they come from inside the neighbour patch and where they land in the LDS
    // copy does not depend on the block -- decoded ONCE per kernel into four integers per register slot (patch slot, face, offset inside the
    // neighbour's halo-less patch, LDS offset); per block only the neighbour patch is looked up.  A unit is two doubles (one 16-byte load) where the
    // variable count is even and the arrays are 16-byte aligned, one double otherwise.  (Decoded per element and block, the remote halos cost 0.36 of
    // the 1.03 ms of the 2^20-patch step: profiles/r04_fv_grid.txt.)
    // (16-byte units need 16-byte aligned arrays: otherwise every block takes the generic path)
    // (... and run-time shapes whose full blocks hold more remote halo units than the HG2 register slots of a lane cover -- 3-D P = 6 H = 1 with 15
    // variables has 3 240 of them, HG2 * NT = 3 072 -- take the generic path too: it has the loop for what the registers do not hold)
    const bool fastpath = (GWC == 1 || ((reinterpret_cast<unsigned long long>(Q) | (cd.bstate ? reinterpret_cast<unsigned long long>(cd.bstate) : 0ull)) & 15) == 0)
                          && (!GRID || (ppb * nA_pp + 2 * per * V) / GWC <= HG2 * NT);
    [[maybe_unused]] int g_pp[HG2], g_face[HG2], g_src[HG2], g_lds[HG2];
    if constexpr (GRID) {
        const int nu = (ppb * nA_pp + 2 * per * V) / GWC;         // units of a full block
#pragma unroll
        for (int r = 0; r < HG2; r++) {
            const int u0 = (int)threadIdx.x + r * NT, u = u0 < nu ? u0 : nu - 1;
            int pp, hv, v, co[3] = {0, 0, 0}, ax;
            remote_elem(ppb, u * GWC, pp, hv, v);
            const int off = halo_place(pp, hv, co, ax);
            const int side = co[ax] < H ? 0 : 1;
            int cg = 0;
#pragma unroll
            for (int b = 0; b < DIM; b++) cg = cg * P + ((b == ax ? (side == 0 ? co[b] + P : co[b] - P) : co[b]) - H);
            g_pp[r] = pp;
            g_face[r] = ax * 2 + side;
            g_src[r] = cg * V + v;
            g_lds[r] = u0 < nu ? off + v : -1;
        }
    }
    [[maybe_unused]] auto unit_src = [&](int r, int half) -> const double* {
        const int* pgc = pgtab + (half * ppb + g_pp[r]) * 3;
        const int ax = g_face[r] >> 1, side = g_face[r] & 1;
        int pg[3] = {pgc[0], pgc[1], pgc[2]};
        if (cd.bstate && (side == 0 ? pg[ax] == 0 : pg[ax] == cd.g[ax] - 1)) return cd.bstate + (long)g_face[r] * V + (g_src[r] % V);
        pg[ax] += side == 0 ? -1 : 1;
        if (pg[ax] < 0) pg[ax] += cd.g[ax];
        if (pg[ax] >= cd.g[ax]) pg[ax] -= cd.g[ax];
        const long np = DIM == 3 ? ((long)pg[0] * cd.g[1] + pg[1]) * cd.g[2] + pg[2] : (long)pg[0] * cd.g[1] + pg[1];
        return Q + np * ncell * V + g_src[r];
    };
    [[maybe_unused]] auto gather_request = [&](long blk, int half) {
        if (!fastpath || n_patches - blk * ppb < ppb) { gather_request_any(blk, half); return; }      // (misaligned arrays; ragged last block)
#pragma unroll
        for (int r = 0; r < HG2; r++) {
            const double* src = unit_src(r, half);
            if constexpr (GWC == 2) {
                const v2d t2 = *reinterpret_cast<const v2d*>(src);
                hg[2 * r] = t2.x;
                hg[2 * r + 1] = t2.y;
            } else {
                hg[r] = *src;
            }
        }
    };
    [[maybe_unused]] auto gather_land = [&](long blk, int half) {
        if (!fastpath || n_patches - blk * ppb < ppb) { gather_land_any(blk, half); return; }
#pragma unroll
        for (int r = 0; r < HG2; r++) {
            if (g_lds[r] >= 0) {
                if constexpr (GWC == 2) *reinterpret_cast<v2d*>(fv_lds + g_lds[r]) = v2d{hg[2 * r], hg[2 * r + 1]};
                else fv_lds[g_lds[r]] = hg[r];
            }
        }
    };
    // faces of the last axis: from the neighbour's interior in the same LDS copy (which the staging copy wrote and nobody overwrites: only halo
    // entries are written here); a neighbour outside the block (a periodic wrap inside a block) from global memory
    [[maybe_unused]] auto gather_local = [&](long blk, int half) {
        const long first = blk * ppb;
        const int npatch = (int)((n_patches - first < ppb) ? n_patches - first : ppb);
        for (int e = (int)threadIdx.x; e < npatch * 2 * per; e += NT) {          // one halo volume per lane: located once, V doubles copied
            const int pp = e / (2 * per), hl = e - pp * (2 * per);   // halo volume among the two faces of the last axis
            if ((pp == 0 && hl < per) || (pp == npatch - 1 && hl >= per)) continue;   // (the two end faces arrived through the registers)
            int co[3] = {0, 0, 0}, ax;
            double* dst = fv_lds + halo_place(pp, 2 * (DIM - 1) * per + hl, co, ax);
            long np;
            int cn[3] = {0, 0, 0};
            const double* src;
            if (fv_grid_locate<DIM>(cd, pgtab + (half * ppb + pp) * 3, co, ax, P, H, np, cn)) src = cd.bstate + np * V;
            else {
                long cl = 0, cg = 0;
#pragma unroll
                for (int b = 0; b < DIM; b++) { cl = cl * S + cn[b]; cg = cg * P + (cn[b] - H); }
                src = (np >= first && np < first + npatch) ? fv_lds + ((np - first) * vol + cl) * V : Q + (np * ncell + cg) * V;
            }
            for (int v = 0; v < V; v++) dst[v] = src[v];
        }
    };
    if constexpr (PERSIST) request(blockIdx.x);
    if constexpr (PERSIST && GRID) {
        fill_pgtab(blockIdx.x, 0);
        __syncthreads();
        gather_request(blockIdx.x, 0);
    }
    [[maybe_unused]] int half = 0;                                // which half of the coordinate table belongs to the current block
    for (long blk = blockIdx.x; blk < nblk; blk += blk_step, half ^= 1) {
    const long patch = blk * ppb + pl;
    // slot (optional): one entry per patch, < 0 = this patch is not in use (exa_fv_time_step_device_masked: the
    // number of patches in use is known on the device only, the launch covers the array's capacity)
    const bool live = pl < ppb && patch < n_patches && (!slot || slot[patch] >= 0);
    const double* Qp = Q + (live ? patch : 0) * gvol * V;        // this thread's patch in HBM: where the stencil reads it ...
    double* Qg = Q + (live ? patch : 0) * gvol * V;              // ... and where its results go (in place; a grid step writes cd.out instead)
    if constexpr (STAGE) {
        const long first = blk * ppb;
        const long npatch = (n_patches - first < ppb) ? n_patches - first : ppb;
        const long total = npatch * gvol * V;                    // doubles in this workgroup's block
        const double* src = Q + first * gvol * V;
        const bool al16 = ((reinterpret_cast<unsigned long long>(src) & 15) == 0);
        if constexpr (GRID && !PERSIST) {
            fill_pgtab(blk, 0);
            __syncthreads();
            gather_request(blk, 0);                               // in flight during the staging copy
        }
        // grid step: the block is halo-less in HBM and lands in the LDS layout with halo (dense_to_lds); V even: a pair is one volume's
        auto land_pair = [&](int xx, v2d val) {
            const int o0 = dense_to_lds(2 * xx);
            if (V % 2 == 0) *reinterpret_cast<v2d*>(fv_lds + o0) = val;
            else { fv_lds[o0] = val.x; fv_lds[dense_to_lds(2 * xx + 1)] = val.y; }
        };
        if constexpr (PERSIST) {
            const int npair = (int)(total / 2);
            v2d* d2 = reinterpret_cast<v2d*>(fv_lds);
#pragma unroll
            for (int r = 0; r < HRS; r++) {
                const int xx = (int)threadIdx.x + r * NT;
                if (xx < npair) {
                    if constexpr (GRID) land_pair(xx, hold[r]);
                    else d2[xx] = hold[r];
                }
            }
            if constexpr (GRID) {
                if (blk + blk_step < nblk) fill_pgtab(blk + blk_step, half ^ 1);
            }
            __syncthreads();
            if (blk + blk_step < nblk) request(blk + blk_step);  // in flight during the update and the write-back below
        } else
        if constexpr (GRID) {
            if (al16) {
                const v2d* s2 = reinterpret_cast<const v2d*>(src);
                for (long i = threadIdx.x; i < total / 2; i += NT) land_pair((int)i, s2[i]);
                if ((total & 1) && threadIdx.x == 0) fv_lds[dense_to_lds((int)total - 1)] = src[total - 1];
            } else {
                for (long i = threadIdx.x; i < total; i += NT) fv_lds[dense_to_lds((int)i)] = src[i];
            }
        } else
        if (al16) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(fv_lds);
            for (long i = threadIdx.x; i < total / 2; i += NT) d2[i] = s2[i];
            if ((total & 1) && threadIdx.x == 0) fv_lds[total - 1] = src[total - 1];
        } else {
            for (long i = threadIdx.x; i < total; i += NT) fv_lds[i] = src[i];
        }
        if constexpr (GRID) {
            // grid step: the face-halo volumes of the LDS copy take the neighbours' interior states (what a halo-fill pass would have
            // written into Q beforehand; here it costs no pass -- the lines are the ones the neighbours' own workgroups read)
            if constexpr (!PERSIST) __syncthreads();             // (the in-block copies read what the staging above wrote; PERSIST: barrier above)
#ifndef EXA_FV_GRID_ABL          // (development: 1 = without the in-block copies, 2 = without the remote halos, 3 = neither -- timing only, wrong results)
#define EXA_FV_GRID_ABL 0
#endif
            if constexpr ((EXA_FV_GRID_ABL & 1) == 0) gather_local(blk, PERSIST ? half : 0);
            if constexpr ((EXA_FV_GRID_ABL & 2) == 0) gather_land(blk, PERSIST ? half : 0);
            if constexpr (PERSIST && (EXA_FV_GRID_ABL & 2) == 0) {
                if (blk + blk_step < nblk) gather_request(blk + blk_step, half ^ 1);
            }
        }
        if constexpr (!PERSIST || GRID) __syncthreads();
        Qp = fv_lds + (long)(live ? pl : 0) * vol * V;
    }
    long st[3];
    if constexpr (DIM == 3) { st[0] = (long)S * S; st[1] = S; st[2] = 1; }
    else { st[0] = S; st[1] = 1; st[2] = 0; }

    double nv[CPT][MAXV];
    long cidx[CPT];
#pragma unroll
    for (int k = 0; k < CPT; k++) {
        const int id = (CPT == 1) ? (int)threadIdx.x - pl * ncell : (int)threadIdx.x + k * NT;
        cidx[k] = -1;
        if (!live || id >= ncell) continue;
#ifdef EXA_FV_ABL_NOCOMPUTE
        {
            int co[3];
            if constexpr (DIM == 3) { co[0] = id / (P * P) + H; co[1] = (id / P) % P + H; co[2] = id % P + H; }
            else { co[0] = id / P + H; co[1] = id % P + H; co[2] = 0; }
            cidx[k] = co[0] * st[0] + co[1] * st[1] + co[2] * st[2];
            for (int v = 0; v < MAXV; v++) nv[k][v] = v < m ? Qp[cidx[k] * V + v] : 0.0;
            continue;
        }
#endif
        int co[3];
        if constexpr (DIM == 3) { co[0] = id / (P * P) + H; co[1] = (id / P) % P + H; co[2] = id % P + H; }
        else { co[0] = id / P + H; co[1] = id % P + H; co[2] = 0; }
        // index of the volume where the stencil reads it: the layout with halo (HBM in place, or the LDS copy); a grid step without an LDS copy
        // reads the halo-less array, where a patch's volumes are numbered like the threads
        constexpr bool BARE = !STAGE && GRID;
        const long c = BARE ? (long)id : co[0] * st[0] + co[1] * st[1] + co[2] * st[2];
        cidx[k] = c;
        // state of the volume next to c along d (BARE: across a patch face it lives in the neighbour patch)
        auto nbr = [&](int d, int sgn) -> const double* {
            if constexpr (BARE) {
                if (sgn < 0 ? co[d] - 1 < H : co[d] + 1 >= P + H) {
                    int cn[3] = {co[0], co[1], co[2]}, pgc[3];
                    cn[d] += sgn;
                    fv_grid_coords<DIM>(cd, patch, pgc);
                    return fv_grid_source<DIM>(Q, cd, pgc, cn, d, P, H, V);
                }
                const long ds = DIM == 3 ? (d == 0 ? (long)P * P : (d == 1 ? P : 1)) : (d == 0 ? P : 1);
                return Qp + (c + sgn * ds) * V;
            }
            return Qp + (c + sgn * st[d]) * V;
        };
        // centre of the volume (exahype2::fv::getVolumeCentre: patch centre - half the patch + (index + 1/2) h) and of its 2 dim neighbours
        [[maybe_unused]] double xc[3] = {0.0, 0.0, 0.0};
        if constexpr (pde_has_xt<PDE>::value) {
#pragma unroll
            for (int a = 0; a < DIM; a++) xc[a] = (cd.centre ? cd.centre[patch * DIM + a] : 0.0) + (co[a] - H + 0.5 - 0.5 * P) * cd.h;
        }
        auto shifted = [&](int d, double sgn, double (&xs)[3]) {
#pragma unroll
            for (int a = 0; a < 3; a++) xs[a] = xc[a] + (a == d ? sgn * cd.h : 0.0);
        };
        double qc[MAXV];
#pragma unroll
        for (int v = 0; v < MAXV; v++) qc[v] = v < m ? Qp[c * V + v] : 0.0;

        if constexpr (MODE == 0) {
            // L6/L7 (test.cpp:60-77): Q_copy = Q_copy - 0.5*F[c+e] + 0.5*F[c-e]; flux rows outside
            // the normal-direction interior were never computed by L2/L3 (test.cpp:22-23) -> zero.
            double acc[MAXV];
#pragma unroll
            for (int v = 0; v < MAXV; v++) acc[v] = qc[v];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double Fp[MAXV], Fm[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) { Fp[v] = 0.0; Fm[v] = 0.0; }
                double qP[MAXV], qM[MAXV];
                const double *pP = nbr(d, 1), *pM = nbr(d, -1);
#pragma unroll
                for (int v = 0; v < MAXV; v++) {
                    qP[v] = v < m ? pP[v] : 0.0;
                    qM[v] = v < m ? pM[v] : 0.0;
                }
                [[maybe_unused]] double xp[3], xm[3];
                if constexpr (pde_has_xt<PDE>::value) { shifted(d, 1.0, xp); shifted(d, -1.0, xm); }
                if (co[d] + 1 < P + H) fv_flux<PDE>(qP, xp, cd.t, d, Fp);
                if (co[d] - 1 >= H) fv_flux<PDE>(qM, xm, cd.t, d, Fm);
#pragma unroll
                for (int v = 0; v < MAXV; v++) acc[v] = acc[v] - 0.5 * Fp[v] + 0.5 * Fm[v];
            }
            // L8/L9 (test.cpp:78-95): variable 0 only, reads the ORIGINAL Q
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double qP[MAXV], qM[MAXV];
                const double *pP = nbr(d, 1), *pM = nbr(d, -1);
#pragma unroll
                for (int v = 0; v < MAXV; v++) {
                    qP[v] = v < m ? pP[v] : 0.0;
                    qM[v] = v < m ? pM[v] : 0.0;
                }
                [[maybe_unused]] double xp[3], xm[3];
                if constexpr (pde_has_xt<PDE>::value) { shifted(d, 1.0, xp); shifted(d, -1.0, xm); }
                const double lc = fv_eig<PDE>(qc, xc, cd.t, d);
                const double lp = (co[d] + 1 < P + H) ? fv_eig<PDE>(qP, xp, cd.t, d) : 0.0;
                const double lm = (co[d] - 1 >= H) ? fv_eig<PDE>(qM, xm, cd.t, d) : 0.0;
                const double mp = lp > lc ? lp : lc;     // Functions.cpp:64-66 std::max(*a,*b)
                const double mm = lm > lc ? lm : lc;
                const double qp0 = qP[0], qm0 = qM[0], q0 = qc[0];
                acc[0] = 0.5 * dt * ((-qp0 + q0) * mp + (qm0 - q0) * mm) + acc[0];
            }
#pragma unroll
            for (int v = 0; v < MAXV; v++) nv[k][v] = acc[v];
        } else {
            double acc[MAXV];
#pragma unroll
            for (int v = 0; v < MAXV; v++) acc[v] = 0.0;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double qpp[MAXV], qmp[MAXV];
                const double *pP = nbr(d, 1), *pM = nbr(d, -1);
#pragma unroll
                for (int v = 0; v < MAXV; v++) {
                    qpp[v] = v < m ? pP[v] : 0.0;
                    qmp[v] = v < m ? pM[v] : 0.0;
                }
                const double* qcp = qc;
                [[maybe_unused]] double xp[3], xm[3];
                if constexpr (pde_has_xt<PDE>::value) { shifted(d, 1.0, xp); shifted(d, -1.0, xm); }
                const double lc = fv_eig<PDE>(qcp, xc, cd.t, d);
                const double sp = fmax(lc, fv_eig<PDE>(qpp, xp, cd.t, d));
                const double sm = fmax(fv_eig<PDE>(qmp, xm, cd.t, d), lc);
                double Fc[MAXV], Fn[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) { Fc[v] = 0.0; Fn[v] = 0.0; }
                fv_flux<PDE>(qcp, xc, cd.t, d, Fc);
                fv_flux<PDE>(qpp, xp, cd.t, d, Fn);
#pragma unroll
                for (int v = 0; v < MAXV; v++)
                    if (v < m) acc[v] += 0.5 * (Fc[v] + Fn[v]) - 0.5 * sp * (qpp[v] - qc[v]);
#pragma unroll
                for (int v = 0; v < MAXV; v++) Fn[v] = 0.0;
                fv_flux<PDE>(qmp, xm, cd.t, d, Fn);
#pragma unroll
                for (int v = 0; v < MAXV; v++)
                    if (v < m) acc[v] -= 0.5 * (Fn[v] + Fc[v]) - 0.5 * sm * (qc[v] - qmp[v]);
                if constexpr (pde_has_ncp<PDE>::value) {             // + (D_{c,c+1} + D_{c-1,c}) / 2, each at the face's mean state (and mid point)
                    double qa[MAXV], dq[MAXV], D[MAXV];
                    [[maybe_unused]] double xf[3];
#pragma unroll
                    for (int side = 0; side < 2; side++) {
#pragma unroll
                        for (int v = 0; v < MAXV; v++) {
                            const double qo = side == 0 ? qpp[v] : qmp[v];
                            qa[v] = 0.5 * (qc[v] + qo);
                            dq[v] = side == 0 ? qo - qc[v] : qc[v] - qo;
                            D[v] = 0.0;
                        }
                        if constexpr (pde_has_xt<PDE>::value) shifted(d, side == 0 ? 0.5 : -0.5, xf);
                        fv_ncp<PDE>(qa, dq, xf, cd.t, d, D);
#pragma unroll
                        for (int v = 0; v < MAXV; v++)
                            if (v < m) acc[v] += 0.5 * D[v];
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < MAXV; v++) nv[k][v] = qc[v] - dt_over_h * acc[v];
            if constexpr (pde_has_source<PDE>::value) {              // q_t + div F = S(q): + dt S(q) of the volume itself
                double Sq[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) Sq[v] = 0.0;
                fv_source<PDE>(qc, xc, cd.t, Sq);
#pragma unroll
                for (int v = 0; v < MAXV; v++)
                    if (v < m) nv[k][v] += dt * Sq[v];
            }
        }
        if constexpr (GRID) {                                        // the next step's CFL scan: eigenvalues of the NEW state of this volume
            if (cd.lam) {
#pragma unroll
                for (int d = 0; d < DIM; d++) lmax = nan_max(lmax, fv_eig<PDE>(nv[k], xc, cd.t + dt, d));
            }
        }
    }
    // staged + (auxiliary variables present or out of place), V even, arrays 16-byte aligned: results go into the LDS copy first, then whole
    // interior rows (P*V contiguous doubles) stream out with coalesced 16-byte stores -- back into Q, or packed densely into QOut
    bool rows = false;
    if constexpr (STAGE)
        rows = ((reinterpret_cast<unsigned long long>(Q) & 15) == 0) && (V % 2 == 0) &&
               (cd.out ? (reinterpret_cast<unsigned long long>(cd.out) & 15) == 0 : V > m);
    if (cd.out && !rows) {
        // out of place (the CellData flavour): QOut is halo-less, [patch][volume][n_real + n_aux] -- thread order is volume order, every
        // thread writes the V contiguous doubles of its volume (evolved variables updated, auxiliary ones copied): fully coalesced
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            if (cidx[k] < 0) continue;
            const int id = (CPT == 1) ? (int)threadIdx.x - pl * ncell : (int)threadIdx.x + k * NT;
            double* o = cd.out + (patch * ncell + id) * V;
#pragma unroll
            for (int v = 0; v < MAXV; v++)
                if (v < m) o[v] = nv[k][v];
            for (int v = m; v < V; v++) o[v] = Qp[cidx[k] * V + v];
        }
        if constexpr (PERSIST) __syncthreads();                  // the LDS copy is free for the next block
        continue;
    }
    // every read of this patch is done (loads feed the values above) before any write
    __syncthreads();
#ifdef EXA_FV_ABL_NOWRITE
    if (rows) { if (nv[0][0] == 1.2345e-300) Q[0] = 0.0; rows = false; if constexpr (PERSIST) __syncthreads(); continue; }
#endif
    if (rows) {
        double* Ql = fv_lds + (long)(live ? pl : 0) * vol * V;
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            if (cidx[k] < 0) continue;
#pragma unroll
            for (int v = 0; v < MAXV; v++)
                if (v < m) Ql[cidx[k] * V + v] = nv[k][v];
        }
        __syncthreads();
        const long first = blk * ppb;
        const long npatch = (n_patches - first < ppb) ? n_patches - first : ppb;
        const int rows_pp = (DIM == 3) ? P * P : P;                  // interior rows per patch
        const int r2 = P * V / 2;                                    // double2 per row
        double* dst = Q + first * vol * V;
        if (cd.out) {                                                // QOut: [patch][interior row][P*V], dense
            v2d* od = reinterpret_cast<v2d*>(cd.out + first * ncell * V);
            for (long e = threadIdx.x; e < npatch * rows_pp * r2; e += NT) {
                const long row = e / r2;
                const int x = (int)(e - row * r2);
                const long pp = row / rows_pp;
                const int rr = (int)(row - pp * rows_pp);
                const long o = pp * vol * V + ((DIM == 3) ? ((long)(rr / P + H) * S * S + (long)(rr % P + H) * S + H) : ((long)(rr + H) * S + H)) * V;
                __builtin_nontemporal_store(reinterpret_cast<const v2d*>(fv_lds + o)[x], od + e);
            }
            if constexpr (PERSIST) __syncthreads();
            continue;
        }
        // (tried in r2: rows j in [H, P+H) with ALL k as one contiguous block per patch, the k-halo volumes rewriting their old
        // values -- no partial lines but +50 % bytes written at P = 4: 1.07 ms against 1.00 ms; the kernel moves its ACTUAL
        // 4.5 GB at 4.5 TB/s either way)
#ifdef EXA_FV_WALIGN
        // whole EXA_FV_WALIGN-byte units: a row's span is widened to the unit boundaries either side, the halo bytes in those units
        // rewrite their own (unchanged) values from the LDS copy -- no partially written unit reaches the memory controller
        constexpr int AL = EXA_FV_WALIGN / 8;
        const int mis = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & (AL - 1));
        const int slots = (P * V + 2 * (AL - 1) + 1) / 2;
        for (long e = threadIdx.x; e < npatch * rows_pp * slots; e += NT) {
            const long row = e / slots;
            const int x = (int)(e - row * slots);
            const long pp = row / rows_pp;
            const int rr = (int)(row - pp * rows_pp);
            const long o = pp * vol * V + ((DIM == 3) ? ((long)(rr / P + H) * S * S + (long)(rr % P + H) * S + H) : ((long)(rr + H) * S + H)) * V;
            long lo = ((o + mis) & ~(long)(AL - 1)) - mis, hi = ((o + P * V + mis + AL - 1) & ~(long)(AL - 1)) - mis;
            lo = lo < pp * vol * V ? pp * vol * V : lo;
            hi = hi > (pp + 1) * vol * V ? (pp + 1) * vol * V : hi;
            const long at = lo + 2 * x;
            if (at < hi) __builtin_nontemporal_store(*reinterpret_cast<const v2d*>(fv_lds + at), reinterpret_cast<v2d*>(dst + at));
        }
#else
        for (long e = threadIdx.x; e < npatch * rows_pp * r2; e += NT) {
            const long row = e / r2;
            const int x = (int)(e - row * r2);
            const long pp = row / rows_pp;
            const int rr = (int)(row - pp * rows_pp);
            const long o = pp * vol * V + ((DIM == 3) ? ((long)(rr / P + H) * S * S + (long)(rr % P + H) * S + H) : ((long)(rr + H) * S + H)) * V;
            // streaming store: the rows start at 80-byte offsets, i.e. partial cache lines -- do not allocate them in L2
            fv_row_store(reinterpret_cast<const v2d*>(fv_lds + o)[x], reinterpret_cast<v2d*>(dst + o) + x);
        }
#endif
    } else {
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            if (cidx[k] < 0) continue;
#pragma unroll
            for (int v = 0; v < MAXV; v++)
                if (v < m) Qg[cidx[k] * V + v] = nv[k][v];
        }
    }
    if constexpr (PERSIST) __syncthreads();                      // the LDS copy is free for the next block
    }
    if constexpr (GRID) {
        if (cd.lam) fv_lam_commit(cd.lam, lmax);
    }
}

// Large 3-D patches (cfg 4's limiter patch: 17^3 x 5 doubles = 196 KB with halo, more than LDS): the patch streams
// plane by plane (axis 0) through a 3-slot LDS ring; thread (j, k) updates the volume (i, j, k) of the current plane
// and keeps its own column (i-1, i, i+1) in registers, so the ring only has to hold planes i and i+1 plus the slot
// plane i+2 lands in: 35 KB per workgroup (49 KB with the cached scalars) -> three workgroups per CU.
//
// Data movement (what bounds the kernel): plane i+2 is fetched into REGISTERS at the top of iteration i (16-byte
// global loads wherever the 8-byte-aligned plane allows: planes of S*S*V = 1445 doubles alternate between 16-byte
// aligned and not) and written to its ring slot during the update of plane i, so a whole iteration hides its latency.
// Results do not go back volume by volume (V = 5 doubles at a 40-byte stride: partial lines): once every stencil read
// of plane i is done (barrier) they overwrite plane i's ring slot, whose rows j in [H, P+H) are ONE contiguous block
// of P*S*V doubles in HBM (the k-halo volumes of a row keep their old values) and stream out with coalesced 16-byte
// stores.
//
// Arithmetic (CACHE: corrected mode of a PDE with the fv_aux interface): 1/rho, p and the sound speed are computed
// once per volume when its plane arrives and kept in a 2-plane LDS ring beside the state -- the seven stencil visits
// of a volume no longer repeat two IEEE divisions and a square root each (13 divisions + 7 square roots per volume
// before) -- and the x-face flux of the previous plane is reused.  Faithful mode keeps the reference's arithmetic
// untouched (bit-exact), only the data movement is shared.
template <class P, class = void> struct has_fv_cache : std::false_type {};
template <class P> struct has_fv_cache<P, std::void_t<decltype(P::NFVAUX)>> : std::true_type {};

constexpr int SLAB_NT = 256;
constexpr int SLAB_NR = 4;          // double2 per thread that hold a plane in flight (<= 2048 doubles per plane)
// ring slot: an even number of doubles with one to spare -- a plane sits in its slot at the parity of its HBM address, so that every
// 16-byte aligned pair of HBM is a 16-byte aligned pair of LDS (ds_write_b128 / ds_read_b128 for the landing and the row blocks)
__host__ __device__ constexpr int slab_slot(int S, int V) { return (S * S * V + 2) & ~1; }
__host__ __device__ constexpr size_t slab_lds_bytes(int S, int V, bool cache) {
    return ((size_t)3 * slab_slot(S, V) + (cache ? (size_t)2 * S * S * 3 : 0)) * sizeof(double);
}

// FITNV: the patch evolves exactly PDE::NV variables (checked by the dispatch): arrays and loops are sized for them, not for MAXV
// GRID (exa_fv_grid_step_device): the patches are the cells of a Cartesian grid and both arrays are HALO-LESS ([patch][P^3][V]).  The LDS ring keeps
// its layout with halo (what the stencil and the scalars' code read): a plane of P^2 V doubles lands row by row inside its S^2 V slot, planes
// H - 1 and P + H ARE the last / first interior planes of the neighbours along axis 0, the face-halo rows and columns of the interior planes are
// requested from the neighbours along axes 1 and 2 together with the plane; the new plane leaves as ONE dense block of P^2 V doubles.  No
// halo-fill pass, no halo bytes in HBM at all: 2 P^3 V doubles per patch and step + the neighbours' boundary layers (which their own
// workgroups read anyway).
constexpr int SLAB_NH = 2;          // doubles per thread that hold the in-plane face-halo values of a plane in flight (grid step: 4 H P V <= 512)
template <class PDE, int MODE, bool CACHE, bool FITNV = false, bool GRID = false>
__global__ void __launch_bounds__(SLAB_NT, 2)
fv_rusanov_slab_kernel(double* __restrict__ Q, int P, int H, int m, int V, double dt, double dt_over_h,
                       const long* __restrict__ slot, FvCellData cd) {
    extern __shared__ __attribute__((aligned(16))) double ring[];
    // grid step: workgroup b of a launch runs on XCD b % 8 -- each XCD takes a CONTIGUOUS range of patches, so that the boundary layers a patch reads
    // from its y / z neighbours (40-byte pieces of 128-byte lines along z) are lines the same L2 holds for the neighbour's own sweep, in flight at the
    // same time (r4: 1.32 x the algorithmic bytes left the L2s at 15^3 patches with the round-robin order)
    const long patch = GRID ? fv_xcd_contiguous(blockIdx.x, gridDim.x) : (long)blockIdx.x;
    if (slot && slot[patch] < 0) return;                           // patch not in use (workgroup-uniform)
    constexpr int NA = CACHE ? 3 : 1;
    constexpr int NQ = (CACHE || FITNV) ? PDE::NV : MAXV;                    // state entries kept per volume (cached variant: m == NV)
    const int S = P + 2 * H;
    const int plane = S * S * V;                                  // doubles per plane
    const int aplane = S * S * NA;
    const int slotd = slab_slot(S, V);
    double* auxr = ring + 3 * slotd;                              // [2][S*S][nz(NA)]   (CACHE only)
    const int orow = S * V;
    const int dplane = P * P * V;                                 // grid step: doubles of a halo-less plane
    double* Qp = Q + patch * (GRID ? (long)P * dplane : (long)S * plane);
    [[maybe_unused]] double* Qo = GRID ? cd.out + patch * P * dplane : Qp;
    const int tid = threadIdx.x;
    // rows of the plane at a pitch of 16 lanes (P <= 16): a 32-lane group holds two whole rows, whose volume indices are distinct
    // mod 32 -- with V odd every per-volume LDS read of the group is conflict-free (rows packed at a pitch of P wrap around: 2 lanes
    // of 32 collide and the instruction takes twice as long; 43 % of the LDS cycles were such conflicts)
    const int j = (tid >> 4) + H, k = (tid & 15) + H;
    const bool cell_ok = (tid >> 4) < P && (tid & 15) < P;
    // volume index inside a plane; idle lanes take the first volume of their 32-lane group (same address as an active lane: a
    // broadcast, not a bank conflict)
    const int x = cell_ok ? j * S + k : ((((tid >> 5) << 1) < P ? ((tid >> 5) << 1) : 0) + H) * S + H;
    const int par0 = (int)((reinterpret_cast<unsigned long long>(Qp) >> 3) & 1);
    // grid step: where plane i comes from -- this patch, or (i outside the interior planes) the neighbour along axis 0; a domain face of a
    // non-periodic grid has no plane: its ring slot is filled with the prescribed state
    [[maybe_unused]] const double* nb_lo = Qp;
    [[maybe_unused]] const double* nb_hi = Qp;
    [[maybe_unused]] bool bnd_lo = false, bnd_hi = false;
    if constexpr (GRID) {
        const long b = patch;
        const int g0 = (int)(b / ((long)cd.g[1] * cd.g[2]));
        const long rest = b - (long)g0 * cd.g[1] * cd.g[2];
        bnd_lo = cd.bstate && g0 == 0;
        bnd_hi = cd.bstate && g0 == cd.g[0] - 1;
        const int gl = g0 == 0 ? cd.g[0] - 1 : g0 - 1, gh = g0 == cd.g[0] - 1 ? 0 : g0 + 1;
        nb_lo = Q + ((long)gl * cd.g[1] * cd.g[2] + rest) * P * dplane;
        nb_hi = Q + ((long)gh * cd.g[1] * cd.g[2] + rest) * P * dplane;
    }
    // plane i (index WITH halo) in HBM; grid step: halo-less planes, the ones beyond the patch from the neighbours' interiors (a boundary
    // plane has no source: any valid address, its ring slot is filled with the state)
    auto plane_src = [&](int i) -> const double* {
        if constexpr (GRID) {
            if (i < H) return bnd_lo ? Qp : nb_lo + (long)(i + P - H) * dplane;
            if (i >= P + H) return bnd_hi ? Qp : nb_hi + (long)(i - P - H) * dplane;
            return Qp + (long)(i - H) * dplane;
        }
        return Qp + (long)i * plane;
    };
    // first double of the plane in HBM not 16-byte aligned?
    auto spar = [&](int i) -> int { return GRID ? (int)((reinterpret_cast<unsigned long long>(plane_src(i)) >> 3) & 1) : ((par0 + i * plane) & 1); };
    // plane i in the ring: in place at the parity of its HBM address (a straight copy of 16-byte pairs); grid step: at the slot's start
    auto org = [&](int i) -> double* { return ring + (i % 3) * slotd + (GRID ? 0 : spar(i)); };
    // grid step: double e of a halo-less plane -> its place in the plane with halo
    // (row of e = e / (P V) through a float multiply: e < 2048 and the half keeps the product off the row boundaries by 0.5 / (P V), four orders of
    // magnitude above the float rounding -- three instructions instead of the ~20 of an integer division, twelve times per lane and plane)
    [[maybe_unused]] const int prow = P * V;
    [[maybe_unused]] const float inv_prow = 1.0f / (float)prow;
    [[maybe_unused]] auto d2r = [&](int e) -> int { return e + (int)(((float)e + 0.5f) * inv_prow) * (2 * H * V) + (H * S + H) * V; };

    // ---- plane i -> registers (issue) / registers -> ring slot (land)
    v2d hold[SLAB_NR];
    double hold_head = 0.0, hold_tail = 0.0;
    auto issue_to = [&](int i, v2d (&hd)[SLAB_NR], double& hh, double& ht) {
        const double* src = plane_src(i);
        const int n = GRID ? dplane : plane;
        const int head = spar(i);                                 // first double not 16-byte aligned
        const int npair = (n - head) >> 1;
        const v2d* s2 = reinterpret_cast<const v2d*>(src + head);
        // unconditional loads at clamped (in-bounds) indices: a load under a lane condition becomes a branch with its own
        // s_waitcnt vmcnt(0) (and here a scratch spill) -- the requests would go out one at a time
#pragma unroll
        for (int r = 0; r < SLAB_NR; r++) {
            const int xx = tid + r * SLAB_NT;
            hd[r] = s2[xx < npair ? xx : npair - 1];
        }
        hh = src[0];
        ht = src[n - 1];
    };
    auto land_from = [&](int i, const v2d (&hd)[SLAB_NR], double hh, double ht) {
        const int n = GRID ? dplane : plane;
        const int head = spar(i);
        const int npair = (n - head) >> 1;
        double* dst = org(i);
        if constexpr (GRID) {
            if ((i < H && bnd_lo) || (i >= P + H && bnd_hi)) {     // plane beyond a domain face: the prescribed state in every volume (workgroup-uniform)
                const double* bs = cd.bstate + (i < H ? 0 : 1) * V;
                for (int xx = tid; xx < plane; xx += SLAB_NT) dst[xx] = bs[xx % V];
                return;
            }
#pragma unroll
            for (int r = 0; r < SLAB_NR; r++) {                    // rows of P V doubles into rows of S V: 8-byte stores (a pair may straddle a row end)
                const int xx = tid + r * SLAB_NT;
                if (xx < npair) {
                    dst[d2r(head + 2 * xx)] = hd[r].x;
                    dst[d2r(head + 2 * xx + 1)] = hd[r].y;
                }
            }
            if (tid == 0 && head) dst[d2r(0)] = hh;
            if (tid == 1 && ((n - head) & 1)) dst[d2r(n - 1)] = ht;
            return;
        }
        v2d* d2 = reinterpret_cast<v2d*>(dst + head);             // 16-byte aligned in LDS (slot parity = HBM parity)
#pragma unroll
        for (int r = 0; r < SLAB_NR; r++) {
            const int xx = tid + r * SLAB_NT;
            if (xx < npair) d2[xx] = hd[r];
        }
        if (tid == 0 && head) dst[0] = hh;
        if (tid == 1 && ((plane - head) & 1)) dst[plane - 1] = ht;
    };
    // grid step, interior planes: the 4 H P face-halo volumes (rows j < H, j >= P + H; columns k likewise) come from the neighbours along axes 1 and 2.
    // Which entries a lane fetches is fixed for the patch: source of plane 0 (+ i planes; a boundary state does not move), place in the plane
    [[maybe_unused]] double hval[GRID ? SLAB_NH : 1];
    [[maybe_unused]] const double* hsrc[GRID ? SLAB_NH : 1];
    [[maybe_unused]] int hstep[GRID ? SLAB_NH : 1], hoff[GRID ? SLAB_NH : 1];
    if constexpr (GRID) {
        const int nhe = 4 * H * P * V;
        int pgc[3];
        fv_grid_coords<3>(cd, patch, pgc);
#pragma unroll
        for (int r = 0; r < SLAB_NH; r++) {
            const int e0 = tid + r * SLAB_NT, e = e0 < nhe ? e0 : nhe - 1;
            const int v = e % V, ev = e / V;
            const int f = ev / (H * P), rr = ev - f * (H * P);
            const int layer = rr % H, t = rr / H, a = 1 + (f >> 1);
            const int edge = (f & 1) ? P + H + layer : layer;
            const int co[3] = {H, a == 1 ? edge : t + H, a == 2 ? edge : t + H};          // (first interior plane: + (i - H) halo-less planes)
            const double* src = fv_grid_source<3>(Q, cd, pgc, co, a, P, H, V);
            const bool fixed = cd.bstate && src >= cd.bstate && src < cd.bstate + 6 * V;
            hsrc[r] = src + v;
            hstep[r] = fixed ? 0 : dplane;
            hoff[r] = e0 < nhe ? (co[1] * S + co[2]) * V + v : -1;
        }
    }
    [[maybe_unused]] auto issue_halo = [&](int i) {
        if constexpr (GRID) {
            if (i >= H && i < P + H) {
#pragma unroll
                for (int r = 0; r < SLAB_NH; r++) hval[r] = hsrc[r][(long)(i - H) * hstep[r]];   // (unconditional: idle lanes repeat the last entry)
            }
        }
    };
    [[maybe_unused]] auto land_halo = [&](int i) {
        if constexpr (GRID) {
            if (i >= H && i < P + H) {
                double* dst = org(i);
#pragma unroll
                for (int r = 0; r < SLAB_NH; r++)
                    if (hoff[r] >= 0) dst[hoff[r]] = hval[r];
            }
        }
    };
    auto issue = [&](int i) { issue_to(i, hold, hold_head, hold_tail); issue_halo(i); };
    auto land = [&](int i) { land_from(i, hold, hold_head, hold_tail); };
    // per-volume scalars of plane i (every volume of the plane, halo included: the neighbours read them)
    auto make_aux = [&](int i) {
        if constexpr (CACHE) {
            const double* src = org(i);
            double* dst = auxr + (i & 1) * aplane;
            for (int xx = tid; xx < S * S; xx += SLAB_NT) {
                double q[PDE::NV], a[3];
#pragma unroll
                for (int v = 0; v < PDE::NV; v++) q[v] = src[xx * V + v];
                PDE::fv_aux(q, a);
                dst[xx * 3 + 0] = a[0];
                dst[xx * 3 + 1] = a[1];
                dst[xx * 3 + 2] = a[2];
            }
        }
    };

    // rolling column of this lane: states (and scalars) of (i-1, j, k), (i, j, k); x-face flux between them
    double qm[NQ], qc[NQ], qp[NQ], am[nz(NA)], ac[nz(NA)], Fxl[NQ];
    [[maybe_unused]] double lmax = 0.0;
#pragma unroll
    for (int v = 0; v < NQ; v++) { qm[v] = 0.0; qc[v] = 0.0; qp[v] = 0.0; Fxl[v] = 0.0; }
#pragma unroll
    for (int a = 0; a < NA; a++) { am[a] = 0.0; ac[a] = 0.0; }

    // ---- prologue: plane H-1 -> column registers; planes H, H+1 in the ring; plane H+2 requested inside the loop.
    // The first two planes are requested together (the second through a register set that is dead afterwards).
    {
        v2d h2[SLAB_NR];
        double h2_head, h2_tail;
        issue(H - 1);
        issue_to(H, h2, h2_head, h2_tail);
        land(H - 1);
        land_from(H, h2, h2_head, h2_tail);
    }
    if constexpr (GRID) {                                          // face-halo rows / columns of plane H (once per patch: fetched and landed on the spot)
        __syncthreads();
        issue_halo(H);
        land_halo(H);
    }
    issue(H + 1);
    __syncthreads();
    make_aux(H - 1);
    make_aux(H);
    {
        const double* r0 = org(H - 1) + x * V;
        const double* r1 = org(H) + x * V;
#pragma unroll
        for (int v = 0; v < NQ; v++) { qm[v] = (CACHE || FITNV || v < m) ? r0[v] : 0.0; qc[v] = (CACHE || FITNV || v < m) ? r1[v] : 0.0; }
    }
    __syncthreads();                                               // scalars visible; every lane has read plane H-1
    if constexpr (CACHE) {
#pragma unroll
        for (int a = 0; a < 3; a++) { am[a] = auxr[((H - 1) & 1) * aplane + x * 3 + a]; ac[a] = auxr[(H & 1) * aplane + x * 3 + a]; }
        // x-face flux between planes H-1 and H, times two:  2 F* = (f(L) + f(R)) - s (R - L)   (the 1/2 goes into the final scale;
        // explicit fma: this unit is compiled without contraction for the faithful mode's sake)
        double FL[PDE::NV], FR[PDE::NV];
        PDE::template flux_fv<0>(qm, am, FL);
        PDE::template flux_fv<0>(qc, ac, FR);
        const double sxl = fmax(PDE::template maxeig_fv<0>(qm, am), PDE::template maxeig_fv<0>(qc, ac));
#pragma unroll
        for (int v = 0; v < PDE::NV; v++) Fxl[v] = fma(-sxl, qc[v] - qm[v], FL[v] + FR[v]);
    }
    land(H + 1);                                                   // slot of plane H-2 == slot (H+1) % 3: nothing lives there yet
    __syncthreads();                                               // plane H+1 in the ring (the scalars of H-1 are in registers now)
    if constexpr (GRID) {
        land_halo(H + 1);                                          // (behind the barrier: the landing wrote the same entries with the stale copy)
        __syncthreads();
    }

    for (int i = H; i < P + H; i++) {
        // ---- stage 1: request plane i+2 (stays in registers during the update); scalars of plane i+1
        const bool more = i + 2 <= P + H;                          // planes H-1 .. P+H are all the stencil reads
        if (more) issue(i + 2);
        make_aux(i + 1);                                           // aux slot (i+1)&1 held plane i-1: in registers since iteration i-1
        __syncthreads();                                           // (A) scalars of i+1 visible; the rows of plane i-1 have left LDS
        // ---- stage 2: update volume (i, j, k) in registers
        double out[NQ];
        {
            const double* rc = org(i);
            const double* rp = org(i + 1);
#pragma unroll
            for (int v = 0; v < NQ; v++) qp[v] = (CACHE || FITNV || v < m) ? rp[x * V + v] : 0.0;
            if constexpr (CACHE) {
                const double* ap = auxr + ((i + 1) & 1) * aplane;
                const double* acp = auxr + (i & 1) * aplane;
                double app[3], acc[PDE::NV];
#pragma unroll
                for (int a = 0; a < 3; a++) app[a] = ap[x * 3 + a];
                // x: low face from the previous iteration, high face now (kept for the next one)
                {
                    double FL[PDE::NV], FR[PDE::NV];
                    PDE::template flux_fv<0>(qc, ac, FL);
                    PDE::template flux_fv<0>(qp, app, FR);
                    const double sh = fmax(PDE::template maxeig_fv<0>(qc, ac), PDE::template maxeig_fv<0>(qp, app));
#pragma unroll
                    for (int v = 0; v < PDE::NV; v++) {
                        const double Fh = fma(-sh, qp[v] - qc[v], FL[v] + FR[v]);
                        acc[v] = Fh - Fxl[v];
                        Fxl[v] = Fh;
                    }
                }
                static_for_fv<1, 3>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    const int st = (D == 1) ? S : 1;
                    double qn[PDE::NV], an[3], Fc[PDE::NV], Fn[PDE::NV];
                    PDE::template flux_fv<D>(qc, ac, Fc);
                    const double lc = PDE::template maxeig_fv<D>(qc, ac);
#pragma unroll
                    for (int sgn = 0; sgn < 2; sgn++) {
                        const int xn = sgn ? x + st : x - st;
#pragma unroll
                        for (int v = 0; v < PDE::NV; v++) qn[v] = rc[xn * V + v];
#pragma unroll
                        for (int a = 0; a < 3; a++) an[a] = acp[xn * 3 + a];
                        PDE::template flux_fv<D>(qn, an, Fn);
                        const double sf = fmax(lc, PDE::template maxeig_fv<D>(qn, an));
#pragma unroll
                        for (int v = 0; v < PDE::NV; v++) {
                            if (sgn) acc[v] += fma(-sf, qn[v] - qc[v], Fc[v] + Fn[v]);
                            else acc[v] -= fma(-sf, qc[v] - qn[v], Fn[v] + Fc[v]);
                        }
                    }
                });
#pragma unroll
                for (int v = 0; v < PDE::NV; v++) out[v] = fma(-0.5 * dt_over_h, acc[v], qc[v]);
#pragma unroll
                for (int a = 0; a < 3; a++) { am[a] = ac[a]; ac[a] = app[a]; }
            } else {
                const double* c0 = rc + x * V;
                const int co[3] = {i, j, k};
                // neighbour states: axis 0 from the column registers, axes 1 and 2 from the ring
                double qN[3][2][NQ];
#pragma unroll
                for (int v = 0; v < NQ; v++) {
                    qN[0][0][v] = qm[v];
                    qN[0][1][v] = qp[v];
                    qN[1][0][v] = (FITNV || v < m) ? c0[v - S * V] : 0.0;
                    qN[1][1][v] = (FITNV || v < m) ? c0[v + S * V] : 0.0;
                    qN[2][0][v] = (FITNV || v < m) ? c0[v - V] : 0.0;
                    qN[2][1][v] = (FITNV || v < m) ? c0[v + V] : 0.0;
                }
                if constexpr (MODE == 0) {
                    double acc[NQ];
#pragma unroll
                    for (int v = 0; v < NQ; v++) acc[v] = qc[v];
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        double Fp[NQ], Fm[NQ];
#pragma unroll
                        for (int v = 0; v < NQ; v++) { Fp[v] = 0.0; Fm[v] = 0.0; }
                        if (co[d] + 1 < P + H) PDE::flux_rt(qN[d][1], d, Fp);
                        if (co[d] - 1 >= H) PDE::flux_rt(qN[d][0], d, Fm);
#pragma unroll
                        for (int v = 0; v < NQ; v++) acc[v] = acc[v] - 0.5 * Fp[v] + 0.5 * Fm[v];
                    }
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        const double lc = PDE::maxeig(qc, d);
                        const double lp = (co[d] + 1 < P + H) ? PDE::maxeig(qN[d][1], d) : 0.0;
                        const double lm = (co[d] - 1 >= H) ? PDE::maxeig(qN[d][0], d) : 0.0;
                        const double mp = lp > lc ? lp : lc;
                        const double mm = lm > lc ? lm : lc;
                        acc[0] = 0.5 * dt * ((-qN[d][1][0] + qc[0]) * mp + (qN[d][0][0] - qc[0]) * mm) + acc[0];
                    }
#pragma unroll
                    for (int v = 0; v < NQ; v++) out[v] = acc[v];
                } else {
                    double acc[NQ];
#pragma unroll
                    for (int v = 0; v < NQ; v++) acc[v] = 0.0;
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        double Fc[NQ], Fn[NQ];
#pragma unroll
                        for (int v = 0; v < NQ; v++) { Fc[v] = 0.0; Fn[v] = 0.0; }
                        const double lc = PDE::maxeig(qc, d);
                        const double sp = fmax(lc, PDE::maxeig(qN[d][1], d));
                        const double sm = fmax(PDE::maxeig(qN[d][0], d), lc);
                        PDE::flux_rt(qc, d, Fc);
                        PDE::flux_rt(qN[d][1], d, Fn);
#pragma unroll
                        for (int v = 0; v < NQ; v++)
                            if (FITNV || v < m) acc[v] += 0.5 * (Fc[v] + Fn[v]) - 0.5 * sp * (qN[d][1][v] - qc[v]);
#pragma unroll
                        for (int v = 0; v < NQ; v++) Fn[v] = 0.0;
                        PDE::flux_rt(qN[d][0], d, Fn);
#pragma unroll
                        for (int v = 0; v < NQ; v++)
                            if (FITNV || v < m) acc[v] -= 0.5 * (Fn[v] + Fc[v]) - 0.5 * sm * (qc[v] - qN[d][0][v]);
                    }
#pragma unroll
                    for (int v = 0; v < NQ; v++) out[v] = qc[v] - dt_over_h * acc[v];
                    if constexpr (pde_has_source<PDE>::value) {
                        double Sq[NQ];
#pragma unroll
                        for (int v = 0; v < NQ; v++) Sq[v] = 0.0;
                        PDE::source(qc, Sq);
#pragma unroll
                        for (int v = 0; v < NQ; v++)
                            if (FITNV || v < m) out[v] += dt * Sq[v];
                    }
                }
            }
            // shift the column
#pragma unroll
            for (int v = 0; v < NQ; v++) { qm[v] = qc[v]; qc[v] = qp[v]; }
            if constexpr (GRID) {                                  // the next step's CFL scan: eigenvalues of the NEW state of this volume
                if (cd.lam && cell_ok) {
                    if constexpr (CACHE) {
                        double an[3];
                        PDE::fv_aux(out, an);
                        lmax = nan_max(lmax, nan_max(PDE::template maxeig_fv<0>(out, an), nan_max(PDE::template maxeig_fv<1>(out, an), PDE::template maxeig_fv<2>(out, an))));
                    } else {
#pragma unroll
                        for (int d = 0; d < 3; d++) lmax = nan_max(lmax, PDE::maxeig(out, d));
                    }
                }
            }
        }
        // keep the update above and the landing below (the scheduler otherwise sinks the register-only arithmetic
        // under barrier (B) and waits for plane i+2 right after (A), i.e. before its latency is hidden)
#pragma unroll
        for (int v = 0; v < NQ; v++) asm volatile("" : "+v"(out[v]));        // the results exist HERE (no sinking below the barrier)
        __builtin_amdgcn_sched_barrier(0);
        if (more) land(i + 2);                                     // slot (i+2)%3 held plane i-1: its rows left LDS before (A)
        __syncthreads();                                           // (B) every stencil read of plane i is done
        if constexpr (GRID) {
            if (more) land_halo(i + 2);                            // (behind a barrier after the landing, visible after (C))
        }
        if (cell_ok) {                                             // the evolved variables of (i, j, k) take their new values in place
            double* ob = org(i) + x * V;
#pragma unroll
            for (int v = 0; v < NQ; v++)
                if (CACHE || FITNV || v < m) ob[v] = out[v];
        }
        __syncthreads();                                           // (C) plane i's slot holds the new plane; plane i+2 is in the ring
        // ---- stage 3: rows j in [H, P+H) of plane i -> HBM, one contiguous block, 16-byte stores where aligned
        if constexpr (GRID) {
            // halo-less output: the P x P interior of the plane is ONE dense block of P^2 V doubles (full lines, no halo bytes)
            const double* ob = org(i);
            double* dst = Qo + (long)(i - H) * dplane;
            const int head = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 1);
            const int npair = (dplane - head) >> 1;
            v2d* d2 = reinterpret_cast<v2d*>(dst + head);
            for (int xx = tid; xx < npair; xx += SLAB_NT) d2[xx] = v2d{ob[d2r(head + 2 * xx)], ob[d2r(head + 2 * xx + 1)]};
            if (tid == 0 && head) dst[0] = ob[d2r(0)];
            if (tid == 1 && ((dplane - head) & 1)) dst[dplane - 1] = ob[d2r(dplane - 1)];
        } else {
            const double* ob = org(i) + H * orow;
            double* dst = Qp + (long)i * plane + (long)H * orow;
            const int n = P * orow;
            const int head = (par0 + i * plane + H * orow) & 1;
            const int npair = (n - head) >> 1;
            v2d* d2 = reinterpret_cast<v2d*>(dst + head);
            const v2d* o2 = reinterpret_cast<const v2d*>(ob + head);   // 16-byte aligned in LDS as in HBM
            for (int xx = tid; xx < npair; xx += SLAB_NT) d2[xx] = o2[xx];
            if (tid == 0 && head) dst[0] = ob[0];
            if (tid == 1 && ((n - head) & 1)) dst[n - 1] = ob[n - 1];
        }
    }
    if constexpr (GRID) {
        if (cd.lam) fv_lam_commit(cd.lam, lmax);
    }
}

// CFL scan of a patch array: max over the interior volumes and the directions of the largest eigenvalue (the halo layers are not looked at: a
// grid step never fills them).  Non-negative doubles order like their bit patterns -> integer atomic max into out[0] (zeroed by the launcher).
template <int DIM, class PDE>
__global__ void fv_maxeig_kernel(const double* __restrict__ Q, int P, int H, int V, long n_patches, const double* __restrict__ centre, double t,
                                 double h, double* __restrict__ out) {
    const int S = P + 2 * H;
    const long ncell = DIM == 3 ? (long)P * P * P : (long)P * P, vol = DIM == 3 ? (long)S * S * S : (long)S * S;
    double mx = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_patches * ncell; i += (long)gridDim.x * blockDim.x) {
        const long patch = i / ncell;
        const int id = (int)(i - patch * ncell);
        int co[3];
        if constexpr (DIM == 3) { co[0] = id / (P * P) + H; co[1] = (id / P) % P + H; co[2] = id % P + H; }
        else { co[0] = id / P + H; co[1] = id % P + H; co[2] = 0; }
        const long c = DIM == 3 ? ((long)co[0] * S + co[1]) * S + co[2] : (long)co[0] * S + co[1];
        const double* qp = Q + (patch * vol + c) * V;
        double q[MAXV];
#pragma unroll
        for (int v = 0; v < MAXV; v++) q[v] = v < V ? qp[v] : 0.0;
        [[maybe_unused]] double x[3] = {0.0, 0.0, 0.0};
        if constexpr (pde_has_xt<PDE>::value) {
#pragma unroll
            for (int a = 0; a < DIM; a++) x[a] = (centre ? centre[patch * DIM + a] : 0.0) + (co[a] - H + 0.5 - 0.5 * P) * h;
        }
#pragma unroll
        for (int d = 0; d < DIM; d++) mx = nan_max(mx, fv_eig<PDE>(q, x, t, d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = nan_max(mx, __shfl_xor(mx, o, 64));
    __shared__ double wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) mx = nan_max(mx, wm[w]);
        atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(mx));
    }
}
template <int DIM, class PDE>
static int fv_maxeig(int P, int H, int V, long n_patches, const double* Q, double* lam, hipStream_t s, const double* centre, double t, double h) {
    hipError_t e = hipMemsetAsync(lam, 0, sizeof(double), s);
    if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return -2; }
    const long n = n_patches * (DIM == 3 ? (long)P * P * P : (long)P * P);
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) return 0;
    hipLaunchKernelGGL((fv_maxeig_kernel<DIM, PDE>), dim3((unsigned)nb), dim3(256), 0, s, Q, P, H, V, n_patches, centre, t, h, lam);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("fv maxeig launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

template <class PDE>
__global__ void pde_eval_kernel(int normal, long n, int stride, const double* __restrict__ Q, double* __restrict__ F,
                                double* __restrict__ lam, const double* __restrict__ X, double t) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // X: [n][3] positions of the states for term sets that depend on position / time (null: the origin)
    [[maybe_unused]] double x[3] = {0.0, 0.0, 0.0};
    if constexpr (pde_has_xt<PDE>::value) {
        if (X) { x[0] = X[i * 3]; x[1] = X[i * 3 + 1]; x[2] = X[i * 3 + 2]; }
    }
    if (F) {
        double f[MAXV];
#pragma unroll
        for (int v = 0; v < MAXV; v++) f[v] = 0.0;
        fv_flux<PDE>(&Q[i * stride], x, t, normal, f);
        for (int v = 0; v < PDE::NFLUX && v < stride; v++) F[i * stride + v] = f[v];
    }
    if (lam) lam[i] = fv_eig<PDE>(&Q[i * stride], x, t, normal);
}

template <int DIM, class PDE, int MODE, bool GRID>
static int fv_dispatch(int P, int H, int m, int V, long n_patches, double* Q, double dt, double h, const long* slot, hipStream_t s,
                       const FvCellData& cd) {
    const long ncell = (DIM == 3) ? (long)P * P * P : (long)P * P;
    const double doh = (MODE == 1) ? dt / h : 0.0;
    const int S = P + 2 * H;
    const long pvol = (DIM == 3) ? (long)S * S * S : (long)S * S;
    if (ncell <= 256) {
        const int ppb = (int)(256 / ncell);
        const dim3 grid((unsigned)((n_patches + ppb - 1) / ppb));
        const size_t lds = (size_t)ppb * pvol * V * sizeof(double) + (GRID ? (size_t)ppb * 32 : 0);      // (+ the grid step's coordinate table)
        // persistent form (next block requested into registers during the update of this one): Q 16-byte aligned, an even
        // number of doubles per block, a grid that fills the chip once
        const bool persist = lds <= 64 * 1024 && (!cd.out || (reinterpret_cast<unsigned long long>(cd.out) & 15) == 0) && ((reinterpret_cast<unsigned long long>(Q) & 15) == 0) && (((long)ppb * (GRID ? ncell : pvol) * V) % 2 == 0) &&
                             n_patches >= (long)ppb * 2048;
        auto persist_grid = [&](const void* kern) -> unsigned {
            int per_cu = 0, dev = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
                return 0;
            const long g = (long)per_cu * cus, nb = (n_patches + ppb - 1) / ppb;
            return (unsigned)(g < nb ? g : nb);
        };
        if (DIM == 2 && P == 4 && H == 1 && m == 5 && V == 10) {    // the reference's configuration (Batched_stateless.py:9)
#ifdef EXA_FV_REF_NT
            {
                constexpr int RNT = EXA_FV_REF_NT;
                const int rppb = RNT / 16;
                const size_t rlds = (size_t)rppb * pvol * V * sizeof(double) + (GRID ? (size_t)rppb * 32 : 0);
                auto kr = fv_rusanov_kernel<DIM, PDE, MODE, 1, RNT, true, FvShape<4, 1, 5, 10>, true, GRID>;
                int per_cu = 0, dev = 0, cus = 0;
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kr), RNT, rlds);
                hipGetDevice(&dev);
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                const long g = (long)per_cu * cus, nb = (n_patches + rppb - 1) / rppb;
                if (persist && g > 0 && n_patches >= (long)rppb * 8192) {
                    static bool said = false;
                    if (!said) { fprintf(stderr, "fv ref: NT %d ppb %d per_cu %d\n", RNT, rppb, per_cu); said = true; }
                    hipLaunchKernelGGL(kr, dim3((unsigned)(g < nb ? g : nb)), dim3(RNT), rlds, s, Q, P, H, m, V, dt, doh, n_patches, rppb, slot, cd);
                    hipError_t e = hipGetLastError();
                    if (e != hipSuccess) { set_error("fv_rusanov launch: %s", hipGetErrorString(e)); return -2; }
                    return 0;
                }
            }
#endif
            auto kp = fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, true, FvShape<4, 1, 5, 10>, true, GRID>;
            const unsigned pg = persist ? persist_grid(reinterpret_cast<const void*>(kp)) : 0;
            if (pg > 0) hipLaunchKernelGGL(kp, dim3(pg), dim3(256), lds, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot, cd);
            else hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, true, FvShape<4, 1, 5, 10>, false, GRID>), grid, dim3(256), lds, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot, cd);
        } else if (lds <= 64 * 1024)     // staged: several workgroups per CU keep HBM requests in flight (the persistent form
                                         // with its 64 holding VGPRs lost there: 2-D P = 16 0.55 -> 0.67 ms)
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, true, FvRuntimeShape, false, GRID>), grid, dim3(256), lds, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot, cd);
        else
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, false, FvRuntimeShape, false, GRID>), grid, dim3(256), 0, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot, cd);
    } else if (ncell <= 1024) {
        const size_t lds = (size_t)pvol * V * sizeof(double) + (GRID ? 32 : 0);
        if (lds <= 64 * 1024)
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 1024, true, FvRuntimeShape, false, GRID>), dim3((unsigned)n_patches), dim3(1024), lds, s, Q, P, H, m, V, dt, doh, n_patches, 1, slot, cd);
        else
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 1024, false, FvRuntimeShape, false, GRID>), dim3((unsigned)n_patches), dim3(1024), 0, s, Q, P, H, m, V, dt, doh, n_patches, 1, slot, cd);
    } else if (DIM == 3 && P * P <= 256 && S * S * V <= 2 * SLAB_NR * SLAB_NT && !pde_has_xt<PDE>::value && !pde_has_ncp<PDE>::value &&
               (!cd.out || GRID) && (!GRID || 4 * H * P * V <= SLAB_NH * SLAB_NT)) {
        // plane-streaming variant: 3-plane LDS ring (+ 2 planes of per-volume scalars), one workgroup per patch
        constexpr bool CACHE = (MODE == 1) && has_fv_cache<PDE>::value;
        if (CACHE && m != PDE::NV) { set_error("FV Rusanov: the PDE evolves %d variables, got n_real = %d", PDE::NV, m); return -1; }
        const size_t lds = slab_lds_bytes(S, V, CACHE);
        auto kern = CACHE ? fv_rusanov_slab_kernel<PDE, MODE, CACHE, false, GRID>
                          : (m == PDE::NV ? fv_rusanov_slab_kernel<PDE, MODE, false, true, GRID> : fv_rusanov_slab_kernel<PDE, MODE, false, false, GRID>);
        if (lds > 64 * 1024) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) { set_error("hipFuncSetAttribute(fv slab, %zu B LDS): %s", lds, hipGetErrorString(ea)); return -2; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)n_patches), dim3(SLAB_NT), lds, s, Q, P, H, m, V, dt, doh, slot, cd);
    } else if (ncell <= 4096) {
        hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 4, 1024, false, FvRuntimeShape, false, GRID>), dim3((unsigned)n_patches), dim3(1024), 0, s, Q, P, H, m, V, dt, doh, n_patches, 1, slot, cd);
    }
    else {
        set_error("FV patch with %ld volumes exceeds the 4096 a workgroup keeps in registers", ncell);
        return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("fv_rusanov launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

template <int DIM, class PDE>
static int fv_mode(int mode, int P, int H, int m, int V, long n_patches, double* Q, double dt, double h, const long* slot, hipStream_t s,
                   const FvCellData& cd) {
    if (mode == 0) {
        // the faithful mode IS the reference's statement list (test.cpp:60-95): it has no source and no non-conservative product -- a term
        // set that carries one would be integrated as a different PDE without a word
        if constexpr (pde_has_ncp<PDE>::value || pde_has_source<PDE>::value) {
            set_error("FV faithful mode (the reference's statement list) has no source / ncp term: use EXA_FV_RUSANOV for this term set");
            return -1;
        } else {
            if (cd.grid_on) return fv_dispatch<DIM, PDE, 0, true>(P, H, m, V, n_patches, Q, dt, h, slot, s, cd);
            return fv_dispatch<DIM, PDE, 0, false>(P, H, m, V, n_patches, Q, dt, h, slot, s, cd);
        }
    }
    if constexpr (pde_has_xt<PDE>::value) {
        if (slot && !cd.centre) { set_error("FV Rusanov: the masked patch update of a term set whose terms depend on position / time needs the patch centres (exa_fv_time_step_device_masked_at)"); return -1; }
    }
    if (cd.grid_on) return fv_dispatch<DIM, PDE, 1, true>(P, H, m, V, n_patches, Q, dt, h, slot, s, cd);
    return fv_dispatch<DIM, PDE, 1, false>(P, H, m, V, n_patches, Q, dt, h, slot, s, cd);
}

#ifdef EXA_USER_PDE_HEADER
}  // namespace exa
// user-PDE side library: the same fused kernel instantiated for exa::UserPDE
extern "C" int exa_user_nv() { return exa::UserPDE::NV; }
// bit 0: the terms depend on position / time (HAS_XT), bit 1: the term set carries a non-conservative product (HAS_NCP)
extern "C" int exa_user_pde_flags() { return (exa::pde_has_xt<exa::UserPDE>::value ? 1 : 0) | (exa::pde_has_ncp<exa::UserPDE>::value ? 2 : 0); }
static exa::FvCellData make_cd(double* out, const double* centre, double t, double h, const exa::FvGridArgs* grid) {
    exa::FvCellData cd{out, centre, t, h, nullptr, {1, 1, 1}, nullptr, 0};
    if (grid) { cd.out = grid->out; cd.bstate = grid->bstate; cd.lam = grid->lam; cd.grid_on = 1; for (int a = 0; a < 3; a++) cd.g[a] = grid->g[a]; }
    return cd;
}
extern "C" int exa_user_fv_maxeig(int dim, int P, int H, int n_real, int n_aux, long n_patches, const double* Q, double* lam, void* stream,
                                  const double* centre, double t, double h) {
    using namespace exa;
    if (dim == 2) return fv_maxeig<2, UserPDE>(P, H, n_real + n_aux, n_patches, Q, lam, (hipStream_t)stream, centre, t, h);
    if (dim == 3 && UserPDE::MAXDIM >= 3) return fv_maxeig<3, UserPDE>(P, H, n_real + n_aux, n_patches, Q, lam, (hipStream_t)stream, centre, t, h);
    set_error("user PDE: no FV kernel for dim %d", dim);
    return -1;
}
extern "C" int exa_user_fv_launch(int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, double* Q, double dt,
                                  double h, const long* slot, void* stream, double* out, const double* centre, double t, const void* grid) {
    using namespace exa;
    const FvCellData cd = make_cd(out, centre, t, h, static_cast<const FvGridArgs*>(grid));
    const int V = n_real + n_aux;
    if (n_real > MAXV || n_real < UserPDE::NV) { set_error("user PDE evolves %d variables; n_real = %d", UserPDE::NV, n_real); return -1; }
    if (n_patches <= 0) return 0;
    if (dim == 2) return fv_mode<2, UserPDE>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, (hipStream_t)stream, cd);
    if (dim == 3 && UserPDE::MAXDIM >= 3) return fv_mode<3, UserPDE>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, (hipStream_t)stream, cd);
    set_error("user PDE: no FV kernel for dim %d", dim);
    return -1;
}
extern "C" int exa_user_pde_eval(int normal, long n, int stride, const double* Q, double* F, double* lam, void* stream, const double* X, double t) {
    using namespace exa;
    if (n <= 0) return 0;
    hipLaunchKernelGGL((pde_eval_kernel<UserPDE>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, normal, n, stride, Q, F, lam, X, t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("pde_eval launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
namespace exa {
#else
int fv_maxeig_launch(int dim, int P, int H, int n_real, int n_aux, long n_patches, int pde, const double* Q, double* lam, hipStream_t s,
                     const double* centre, double t, double h) {
    const int V = n_real + n_aux;
    if (pde >= 100) return user_fv_maxeig(pde, dim, P, H, n_real, n_aux, n_patches, Q, lam, s, centre, t, h);
    if (dim == 2) {
        if (pde == 0) return fv_maxeig<2, EulerRef2D>(P, H, V, n_patches, Q, lam, s, centre, t, h);
        if (pde == 1) return fv_maxeig<2, Euler>(P, H, V, n_patches, Q, lam, s, centre, t, h);
        if (pde == 2) return fv_maxeig<2, Advection<MAXV>>(P, H, V, n_patches, Q, lam, s, centre, t, h);
    } else if (dim == 3) {
        if (pde == 1) return fv_maxeig<3, Euler>(P, H, V, n_patches, Q, lam, s, centre, t, h);
        if (pde == 2) return fv_maxeig<3, Advection<MAXV>>(P, H, V, n_patches, Q, lam, s, centre, t, h);
    }
    set_error("FV max eigenvalue: no kernel for dim %d, pde %d", dim, pde);
    return -1;
}

int fv_launch(int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, int pde, double* Q, double dt,
              double h, const long* slot, hipStream_t s, double* out, const double* centre, double t, const FvGridArgs* grid) {
    const int V = n_real + n_aux;
    FvCellData cd{out, centre, t, h, nullptr, {1, 1, 1}, nullptr, 0};
    if (grid) { cd.out = grid->out; cd.bstate = grid->bstate; cd.lam = grid->lam; cd.grid_on = 1; for (int a = 0; a < 3; a++) cd.g[a] = grid->g[a]; }
    if (grid && grid->lam) {
        hipError_t e0 = hipMemsetAsync(grid->lam, 0, sizeof(double), s);
        if (e0 != hipSuccess) { set_error("memset: %s", hipGetErrorString(e0)); return -2; }
    }
    if (n_real > MAXV) { set_error("n_real = %d exceeds %d", n_real, MAXV); return -1; }
    if (n_patches <= 0) return 0;
    if (pde >= 100) return user_fv_launch(pde, mode, dim, P, H, n_real, n_aux, n_patches, Q, dt, h, slot, s, out, centre, t, grid);
    if (dim == 2) {
        if (pde == 0) return fv_mode<2, EulerRef2D>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s, cd);
        if (pde == 1) return fv_mode<2, Euler>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s, cd);
        if (pde == 2) return fv_mode<2, Advection<MAXV>>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s, cd);
    } else if (dim == 3) {
        if (pde == 1) return fv_mode<3, Euler>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s, cd);
        if (pde == 2) return fv_mode<3, Advection<MAXV>>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s, cd);
    }
    set_error("FV Rusanov: no kernel for dim %d, pde %d", dim, pde);
    return -1;
}

int pde_eval_launch(int pde, int normal, long n, int stride, const double* Q, double* F, double* lam, hipStream_t s, const double* X, double t) {
    if (n <= 0) return 0;
    if (pde >= 100) return user_pde_eval(pde, normal, n, stride, Q, F, lam, s, X, t);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (pde == 0) hipLaunchKernelGGL((pde_eval_kernel<EulerRef2D>), grid, dim3(256), 0, s, normal, n, stride, Q, F, lam, X, t);
    else if (pde == 1) hipLaunchKernelGGL((pde_eval_kernel<Euler>), grid, dim3(256), 0, s, normal, n, stride, Q, F, lam, X, t);
    else if (pde == 2) hipLaunchKernelGGL((pde_eval_kernel<Advection<1>>), grid, dim3(256), 0, s, normal, n, stride, Q, F, lam, X, t);
    else { set_error("unknown pde %d", pde); return -1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("pde_eval launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

#endif

}  // namespace exa
