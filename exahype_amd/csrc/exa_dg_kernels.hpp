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
// of a cell lives in LDS as SoA [var][time slab][node] (Geo<>), the 1-D operators
// live in a plan-owned HBM image (DgOps<> + lane tables) and reach the FMAs as SGPR
// operands through scalar loads, all contractions are sum-factorised pencil
// products kept in registers.  HBM traffic per cell is the compulsory one: u read
// once, u* written once, traces written once.  (3-D cells with N = 7, 8 do not fit
// the LDS: exa_dg_stream.hpp; the single-stage 2-D step in one launch: exa_dg_fused.hpp.)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "exa_dg_common.hpp"
#include "exa_pde.hpp"
#include "exa_dg_plain.hpp"

namespace exa {

// Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch): hand each XCD a contiguous range of logical blocks,
// so that the traces a cell reads from its y/z neighbours were just touched by the same XCD's L2.
__device__ inline long xcd_contiguous(long b, long n) {
    const long per = n / 8;
    return b < per * 8 ? (b % 8) * per + b / 8 : b;
}

template <int I, int E, class F> __device__ inline void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// ------------------------------------------------------------------------------------------
// Stage A
// ------------------------------------------------------------------------------------------
// The operator block lives in HBM (plan-owned) and is read through a constant-address-space
// pointer that is re-materialised ("laundered") at the head of every phase: the loads become
// s_load_dwordx* next to their use, every matrix entry is an SGPR operand of v_fma_f64, and the
// compiler cannot hoist all ~150 doubles out of the Picard loop and spill them (v1 spent a
// quarter of its VALU instructions on v_readlane/v_writelane spill traffic).
#define EXA_AS4 __attribute__((address_space(4)))
template <int N> __device__ inline const EXA_AS4 DgOps<N>* ops_here(const void* raw) {
    unsigned long long a = reinterpret_cast<unsigned long long>(raw);
    asm volatile("" : "+s"(a));
    return (const EXA_AS4 DgOps<N>*)a;
}

template <int N> struct DgStepOps;
template <int N> __device__ inline const EXA_AS4 DgStepOps<N>* step_here(const void* raw) {
    unsigned long long a = reinterpret_cast<unsigned long long>(raw);
    asm volatile("" : "+s"(a));
    return (const EXA_AS4 DgStepOps<N>*)a;
}

// K consecutive operator entries -> SGPRs, all requested in one batch and pinned there.  Scalar loads share the
// lgkmcnt counter with the LDS and return out of order, so every wait for one is `s_waitcnt lgkmcnt(0)`: left to
// itself the compiler fetched the rows just in time, between the LDS accesses of a phase -- three full drains per
// pencil task and twelve per time update, each exposing the scalar-cache latency to a kernel with three waves per
// SIMD.  One batch at the head of the phase, before its first LDS load, costs one wait.
template <int K> __device__ inline void sload(const EXA_AS4 double* p, double (&d)[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) d[k] = p[k];
#pragma unroll
    for (int k = 0; k + 3 < K; k += 4) asm volatile("" : "+s"(d[k]), "+s"(d[k + 1]), "+s"(d[k + 2]), "+s"(d[k + 3]));
#pragma unroll
    for (int k = K / 4 * 4; k < K; k++) asm volatile("" : "+s"(d[k]));
}
template <int K> __device__ inline void spin(double (&d)[K]) {     // values that are wave-uniform already (kernel arguments)
#pragma unroll
    for (int k = 0; k + 3 < K; k += 4) asm volatile("" : "+s"(d[k]), "+s"(d[k + 1]), "+s"(d[k + 2]), "+s"(d[k + 3]));
#pragma unroll
    for (int k = K / 4 * 4; k < K; k++) asm volatile("" : "+s"(d[k]));
}

// Operator entries that depend on dt: Tdt = -dt T, Tsdt = -dt Tsum, so that the time update is one FMA chain started
// from u.  They live behind the operator image in HBM (dg_step_ops_kernel rewrites them in front of every stage-A
// launch, stream-ordered) and reach the FMAs through scalar loads like the rest of the image -- as kernel arguments
// the compiler kept all 84 SGPRs live across the kernel and spilled them to VGPR lanes.
template <int N> struct DgStepOps {
    double Tdt[N * N];
    double Tsdt[N];
    double TdtT[N * N];    // TdtT[l][l'] = Tdt[l'][l]: the column of one input level contiguous (exa_dg_reg.hpp)
};
template <int N> __global__ void dg_step_ops_kernel(const DgOps<N>* __restrict__ ops, DgStepOps<N>* __restrict__ so, double dt) {
    const int k = threadIdx.x;
    if (k < N * N) {
        so->Tdt[k] = -dt * ops->T[k];
        so->TdtT[(k % N) * N + k / N] = -dt * ops->T[k];
    }
    if (k < N) so->Tsdt[k] = -dt * ops->Tsum[k];
}

// Opaque copies: values derived from them cannot be hoisted out of a loop (the compiler otherwise precomputes every address
// and predicate of every phase once per kernel and spills them).
__device__ inline int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ inline int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }

// Diagnostic build only (-DEXA_STAMPS): per-phase cycle stamps of wave 0, summed into a debug
// buffer of their own (never an output element); the production kernel executes no stamp.
#ifdef EXA_STAMPS
__device__ unsigned long long g_exa_stamps[48];
__device__ unsigned long long g_exa_wg_span[2 * 1024];     // per workgroup: start and end on the constant-rate clock (load balance)
#define EXA_STAMP(slot)                                                                         \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        unsigned long long t_;                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        acc_[slot] += t_ - t_prev_;                                                             \
        t_prev_ = t_;                                                                           \
    } while (0)
#define EXA_STAMP_INIT()                                                                        \
    unsigned long long t_prev_, acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_exa_wg_span[2 * blockIdx.x] = wall_clock64();  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev_)::"memory")
#define EXA_STAMP_FLUSH()                                                                       \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_exa_wg_span[2 * blockIdx.x + 1] = wall_clock64(); \
        if (bt == 0)                                                                            \
            for (int k_ = 0; k_ < 12; k_++) atomicAdd(&g_exa_stamps[grp * 12 + k_], acc_[k_]);  \
    } while (0)
#else
#define EXA_STAMP(slot) do { } while (0)
#define EXA_STAMP_INIT() do { } while (0)
#define EXA_STAMP_FLUSH() do { } while (0)
#endif

// Diagnostic ablations (never in the product build): EXA_ABL_NOFMA drops the contraction FMAs,
// EXA_ABL_NOLDS replaces the Picard loop's LDS reads/writes by register traffic.
#ifdef EXA_ABL_NOLDS
#define EXA_LD2(i) make_double2(um[0] + (double)(i), um[0])
#define EXA_ST2(i, a, b) asm volatile("" ::"v"(a), "v"(b))
#else
#define EXA_LD2(i) (*reinterpret_cast<const double2*>(&lds[i]))
#define EXA_ST2(i, a, b) *reinterpret_cast<double2*>(&lds[i]) = make_double2((a), (b))
#endif
#ifdef EXA_ABL_NOLDS
#define EXA_LD(i) (um[0] + (double)(i))
#define EXA_ST(i, val) asm volatile("" ::"v"(val))
#else
// volatile: keeps the back-end from pairing two 8-byte loads into ds_read2_b64, which runs at half the LDS rate of
// two ds_read_b64 (8 vs 2 x 2 LDS cycles per wave instruction, MI355X_MICROARCH.md LDS table)
#define EXA_LD(i) (*(const volatile __attribute__((address_space(3))) double*)(&lds[i]))
#define EXA_ST(i, val) *(volatile __attribute__((address_space(3))) double*)(&lds[i]) = (val)
#endif
#ifdef EXA_ABL_NOLDS
#define EXA_ATOMIC_ADD(i, val) asm volatile("" ::"v"(val))
#else
#define EXA_ATOMIC_ADD(i, val) (void)__hip_atomic_fetch_add(&lds[i], (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif
#ifdef EXA_ABL_SKIP_D
#define EXA_ABL_COND_SKIP_D && n_it < 0
#else
#define EXA_ABL_COND_SKIP_D
#endif
// Stage B: the eigenvalues of the trace states by the IEEE sequences (default) or by the fast reciprocal / square root.  Measured r4 (same box, two
// rounds, scripts/quick_bench_stage_b.py): 128^3 p = 5 15.70 / 15.85 ms IEEE against 16.52 / 16.50 fast (the dense variant LOSES 4 %: the kernel is
// bound by its memory latency chain, and the shorter arithmetic changes nothing but the schedule), 64^3 p = 7 4.07 / 4.10 against 3.98 / 4.05, 128^3 p = 3
// 5.96 / 6.09 against 6.05 / 6.04 -- not adopted.
#ifndef EXA_STAGE_B_FAST_EIG
#define EXA_STAGE_B_FAST_EIG 0
#endif
#ifdef EXA_ABL_NOFMA
#define EXA_FMA(acc, a, b) asm volatile("" : "+v"(acc) : "v"(a), "v"(b))
#else
#define EXA_FMA(acc, a, b) acc += (a) * (b)
#endif

// s = M F for a centro-antisymmetric N x N matrix M given in its even-odd packing E (DgOps::DEO / KEO): half the FMAs.
template <int N, class EP> __device__ inline void eo_apply(EP E, const double (&F)[N], double (&s)[N]) {
    constexpr int H = N / 2;
    double P[H > 0 ? H : 1], M[H > 0 ? H : 1];
#pragma unroll
    for (int i = 0; i < H; i++) P[i] = M[i] = 0.0;
    double mid = 0.0;
#pragma unroll
    for (int j = 0; j < H; j++) {
        const double e = F[j] + F[N - 1 - j], o = F[j] - F[N - 1 - j];
#pragma unroll
        for (int i = 0; i < H; i++) {
            P[i] += E[j * N + i] * e;
            M[i] += E[j * N + H + i] * o;
        }
        if constexpr (N % 2 == 1) mid += E[j * N + 2 * H] * o;
    }
    if constexpr (N % 2 == 1) {
#pragma unroll
        for (int i = 0; i < H; i++) P[i] += E[H * N + i] * F[H];
        s[H] = mid;
    }
#pragma unroll
    for (int i = 0; i < H; i++) {
        s[i] = M[i] + P[i];
        s[N - 1 - i] = M[i] - P[i];
    }
}

// Derivative sums of one pencil whose N nodes (NV variables each) are in registers:  s[i][v] = sum_j D[i][j] * (1/dx_D) f_D(q_j)[v],
// even-odd form of the centro-antisymmetric D (E = DgOps::DEO through a laundered constant-address-space pointer).
template <int D, int N, class PDE>
__device__ inline void pencil_sums(const double (&qw)[N][PDE::NV], const EXA_AS4 double* Em, double idx_d, double (&s)[N][PDE::NV]) {
    constexpr int NV = PDE::NV, NA = PDE::NAUX, H = N / 2;
    double P[H > 0 ? H : 1][NV], M[H > 0 ? H : 1][NV], mid[NV];
#pragma unroll
    for (int i = 0; i < H; i++)
#pragma unroll
        for (int v = 0; v < NV; v++) P[i][v] = M[i][v] = 0.0;
#pragma unroll
    for (int v = 0; v < NV; v++) mid[v] = 0.0;
#pragma unroll
    for (int j = 0; j < H; j++) {
        const int jm = N - 1 - j;
        double aa[nz(NA)], ab[nz(NA)], Fa[NV], Fb[NV];
        PDE::aux_fast(qw[j], aa);
        PDE::template flux_scaled<D>(qw[j], aa, idx_d, Fa);
        PDE::aux_fast(qw[jm], ab);
        PDE::template flux_scaled<D>(qw[jm], ab, idx_d, Fb);
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
                EXA_FMA(P[i][v], ea, Fa[v]);
                EXA_FMA(M[i][v], eb, Fb[v]);
            }
        }
        if constexpr (N % 2 == 1) {
            const double em = Em[j * N + 2 * H];
#pragma unroll
            for (int v = 0; v < NV; v++) EXA_FMA(mid[v], em, Fb[v]);
        }
    }
    if constexpr (N % 2 == 1) {
        double aa[nz(NA)], Fa[NV];
        PDE::aux_fast(qw[H], aa);
        PDE::template flux_scaled<D>(qw[H], aa, idx_d, Fa);
#pragma unroll
        for (int i = 0; i < H; i++) {
            const double ec = Em[H * N + i];
#pragma unroll
            for (int v = 0; v < NV; v++) EXA_FMA(P[i][v], ec, Fa[v]);
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
}

template <int DIM, int N, class PDE, int CPB> struct StageA {
    using G = Geo<DIM, N>;
    // operator image in HBM: DgOps<N> | lane -> pencil tables (DIM * GW ints) | DgStepOps<N>
    static constexpr size_t PERM_OFF = (sizeof(DgOps<N>) + 15) / 16 * 16;
    static constexpr int NV = PDE::NV;
    static constexpr int ASZ = NV * G::NTS * G::SL;         // one q-sized array
    static constexpr int CS = 3 * ASZ;                       // doubles per cell image: Q, A, B
    static constexpr size_t LDS_BYTES = (size_t)CPB * CS * sizeof(double);
    static constexpr int TD = CPB * G::NN;                   // pencil tasks per direction (= node tasks)
    static constexpr int WPD = (TD + 63) / 64;               // waves per direction group
    static constexpr int NT = DIM * WPD * 64;                // threads per workgroup
    static constexpr size_t STEP_OFF = (PERM_OFF + sizeof(int) * DIM * WPD * 64 + 15) / 16 * 16;
    static constexpr size_t IMAGE_BYTES = STEP_OFF + sizeof(DgStepOps<N>);
    // which form of the Picard loop (see the kernel): measured per order on 3-D Euler, 64^3 cells (profiles/r02_stage_a_variants.txt)
    //   0 "late":  x group stores S_x over Q after barrier (1), barrier (2), time update
    //   1 "early": every pencil loaded, barrier (R), sums stored at once, barrier (1), time update
    // (a third form -- the x group waits for a count of the y / z waves whose loads are done instead of for a barrier -- timed
    //  like 0 at N = 6 and 2 % better than 1 at N = 5; not kept: see the profile note and commit a7e35b6)
#if defined(EXA_A_EARLY_STORE)
    static constexpr int MODE = 1;
#elif defined(EXA_A_LATE_STORE)
    static constexpr int MODE = 0;
#else
    static constexpr int MODE = (DIM == 3 && (N == 5 || N == 3)) ? 1 : 0;
#endif
};

// Cell image in LDS: three q-sized arrays Q | A | B, each SoA [var][time slab][node].
//
// Thread layout: DIM groups of WPD waves, persistent grid (a workgroup walks over cells and prefetches the next
// cell's u).  In a Picard iteration group d computes the pencils of direction d -- all directions at once, 3 waves per
// SIMD at N = 6 (v4, one wave per SIMD, ran ~6 cycles per VALU and ~8.7 per LDS instruction back to back;
// profiles/r01_stage_a_stamps.txt).  Which lane takes which pencil comes from host-built tables (dg_inst.hip
// OpsImage) that make every LDS wave instruction of the phase bank-conflict-free.  The contraction with D uses its
// even-odd form (half the FMAs).  Only plain, unpaired LDS stores are used for the partial sums (fp64 LDS atomics cost
// ~17 LDS cycles per wave instruction, v5; ds_write2_b64 13 cycles for two values against 6 per ds_write_b64):
// y -> A and z -> B directly; x keeps its sums in registers over the barrier and then writes them over Q, which is dead
// by then.  The time update adds the three, is split by variable over the DIM groups (all waves), and writes the new
// iterate back into Q.  The cached scalars of the flux (Euler: 1/rho, p) are recomputed per pencil node instead of
// stored: the freed 20 KiB are what buys the second sum array.
template <int DIM, int N, class PDE, int CPB>
__global__ void __launch_bounds__((StageA<DIM, N, PDE, CPB>::NT))
dg_stage_a_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, double* __restrict__ trace,
                  long ncells, CellBox box, double dt, double idx0, double idx1, double idx2, int n_it,
                  const void* __restrict__ ops_raw) {
    using G = Geo<DIM, N>;
    using SA = StageA<DIM, N, PDE, CPB>;
    constexpr int NT = SA::NT, TD = SA::TD, GW = SA::WPD * 64;
    constexpr int NV = PDE::NV, NA = PDE::NAUX;
    constexpr int NN = G::NN, NF = G::NF, SL = G::SL, NTS = G::NTS;
    constexpr int ASZ = SA::ASZ, CS = SA::CS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // rows 0, 1: this block's cells | the next block's (ping-pong); rows 2..7: box coordinates of the lane's slot and their step
    // (one static array of 64 * CPB bytes: the dynamic LDS base behind it stays 16-byte aligned for the 16-byte accesses)
    __shared__ long cell_ids[8][CPB];

    const int tid = threadIdx.x;
    EXA_STAMP_INIT();
    const int grp = __builtin_amdgcn_readfirstlane(tid / GW);   // wave-uniform direction group
    const int bt = tid - grp * GW;                               // task index inside the group
    const double idx[3] = {idx0, idx1, idx2};

    // ---- time-update role of this thread: node bt, the variables of its direction group.  The update is split by
    // variable over the DIM groups (all waves take part); the x group, which alone has the post-barrier store of its
    // sums on its path, gets the smallest share (3-D Euler: x 1 variable, y and z 2 each).  Which variables a group owns
    // is a compile-time property of the group (static_for over the groups, wave-uniform branch): straight-line code with
    // constant LDS offsets.
    constexpr int TBASE = NV / DIM, TREM = NV % DIM;
    auto tcnt = [](int g) constexpr { return TBASE + (g >= DIM - TREM ? 1 : 0); };
    auto tv0 = [](int g) constexpr { return g * TBASE + (g > DIM - TREM ? g - (DIM - TREM) : 0); };
    constexpr int NVA = TBASE + (TREM ? 1 : 0);              // largest share
    const bool t_node = bt < TD;
    const int tc = bt / NN, tn = bt - tc * NN;
    const int toff = tc * CS + G::node_off(tn);

    // derivative-phase task of this lane (conflict-free enumeration built on the host: dg_inst.hip OpsImage), -1: none
    const int d_task = grp < DIM ? reinterpret_cast<const int*>(static_cast<const char*>(ops_raw) + SA::PERM_OFF)[grp * GW + bt] : -1;
    const void* step_raw = static_cast<const char*>(ops_raw) + SA::STEP_OFF;

    // Persistent grid: a workgroup walks over blocks of CPB cells (blk, blk + gridDim.x, ...) and fetches the next
    // block's u while it works on this one -- with one workgroup per CU (LDS) nothing else would hide that latency.
    const long nblocks = (box.nbox + CPB - 1) / CPB;
    long blk = blockIdx.x;
    int par = 0;
    // Box slot -> cell id, kept incrementally by the CPB lanes that own a slot: the slot advances by gridDim.x * CPB per
    // block, i.e. by a fixed (sx, sy, sz) in box coordinates with carries -- no 64-bit divisions on the path of wave 0
    // between two cells (box.cell() costs four of them: ~2 K cycles in front of a workgroup barrier).
    // (the state lives in LDS, not in registers: the kernel sits at its VGPR cap and anything live across the cell loop is spilled)
    if (tid < CPB) {
        long b = blk * CPB + tid;
        const long cz = b % box.nb[2];
        const long b1 = b / box.nb[2];
        const long cy = b1 % box.nb[1], cx = b1 / box.nb[1];
        long g = (long)gridDim.x * CPB;
        const long sz = g % box.nb[2];
        g /= box.nb[2];
        cell_ids[2][tid] = cx; cell_ids[3][tid] = cy; cell_ids[4][tid] = cz;
        cell_ids[5][tid] = g / box.nb[1]; cell_ids[6][tid] = g % box.nb[1]; cell_ids[7][tid] = sz;
        cell_ids[0][tid] = b < box.nbox ? ((box.lo[0] + cx) * box.nc[1] + box.lo[1] + cy) * box.nc[2] + box.lo[2] + cz : -1;
    }
    __syncthreads();                                             // cell ids visible
    // u of node (tc, tn), the variables this lane's group updates (run-time base: the group is wave-uniform).  Only these
    // stay in registers over the cell: the kernel sits at its VGPR cap and all five of this cell plus all five of the next
    // one (prefetch) were what got spilled to scratch -- 9 GB of stray writes per 128^3 launch.
    const int tvb = grp * TBASE + (grp > DIM - TREM ? grp - (DIM - TREM) : 0);
    const int tvn = TBASE + (grp >= DIM - TREM ? 1 : 0);
    double um[NVA > 0 ? NVA : 1];
    {
        const long cell = bt < TD ? cell_ids[0][tc] : -1;
#pragma unroll
        for (int vv = 0; vv < NVA; vv++) um[vv] = cell >= 0 ? u_in[(cell * NN + tn) * NV + (vv < tvn ? tvb + vv : tvb)] : 1.0;
    }
    for (; blk < nblocks; blk += gridDim.x, par ^= 1) {
    const long* cell_id = cell_ids[par];
    {
        if (tid < CPB) {                                         // the next block's slot of this lane
            long cx = cell_ids[2][tid], cy = cell_ids[3][tid], cz = cell_ids[4][tid];
            cz += cell_ids[7][tid];
            if (cz >= box.nb[2]) { cz -= box.nb[2]; cy += 1; }
            cy += cell_ids[6][tid];
            if (cy >= box.nb[1]) { cy -= box.nb[1]; cx += 1; }
            cx += cell_ids[5][tid];
            cell_ids[2][tid] = cx; cell_ids[3][tid] = cy; cell_ids[4][tid] = cz;
            // (cx >= nb[0] <=> the slot is past the end of the box)
            cell_ids[par ^ 1][tid] = cx < box.nb[0] ? ((box.lo[0] + cx) * box.nc[1] + box.lo[1] + cy) * box.nc[2] + box.lo[2] + cz : -1;
        }
        if (t_node && n_it > 0) {                                // q_0 := u; level 0 only: iteration 0 reads nothing else, its update writes all
            static_for<0, DIM>([&](auto gc) {
                constexpr int GI = decltype(gc)::value;
                if (grp == GI) {
#pragma unroll
                    for (int vv = 0; vv < tcnt(GI); vv++) lds[toff + ((tv0(GI) + vv) * NTS + 0) * SL] = um[vv];
                }
            });
        }
    }
    __syncthreads();
    double unm[NVA > 0 ? NVA : 1];                               // the same of the next block's node, in flight during this block
    {
        const long cn = bt < TD ? cell_ids[par ^ 1][tc] : -1;
#pragma unroll
        for (int vv = 0; vv < NVA; vv++) unm[vv] = cn >= 0 ? u_in[(cn * NN + tn) * NV + (vv < tvn ? tvb + vv : tvb)] : 1.0;
    }
    EXA_STAMP(0);

    // ---- Picard iterations (A.2):  q <- u - dt * T * sum_d (1/dx_d) D_d f_d(q)
    for (int it = 0; it < n_it; it++) {
    if constexpr (SA::MODE == 1) {
    constexpr int W0 = (CPB * NF + 63) / 64;                     // iteration 0: the l = 0 pencils, packed on W0 waves per direction
    const int wave = tid >> 6;
    // Form with an early "reads done" barrier: every lane loads its whole pencil, barrier (R), then computes and stores its
    // sums at once -- x over Q (nobody reads Q any more), y -> A, z -> B.  The serial phase of the other form (x group stores
    // after barrier (1) while the others wait, then barrier (2)) is gone; its price is that no flux is evaluated before the
    // last pencil load of the workgroup has landed.  Measured per order (profiles/r02_stage_a_variants.txt): 8 % faster at
    // N = 5, 4 % at N = 3, 7 % slower at N = 6, even at N = 4 -- hence StageA::MODE.
        static_for<0, DIM>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            const bool wave_mine = it > 0 ? grp == D : (wave >= D * W0 && wave < (D + 1) * W0);      // wave-uniform
            if (wave_mine) {
                constexpr int ps = G::pstride(D);
                constexpr bool WIDE = (ps == 1) && (N % 2 == 0);
                const int k0 = (wave - D * W0) * 64 + (tid & 63);
                const bool mine = it > 0 ? d_task >= 0 : k0 < CPB * NF;
                const int c = it > 0 ? d_task / NN : k0 / NF;
                const int r = it > 0 ? d_task - c * NN : k0 - c * NF;
                const int l = it > 0 ? r / NF : 0, t = it > 0 ? r - l * NF : r;
                const int off = mine ? c * CS + l * SL + G::pbase(D, t) : 0;
                double qw[N][NV];
                if (mine) {
                    if constexpr (WIDE) {
#pragma unroll
                        for (int v = 0; v < NV; v++)
#pragma unroll
                            for (int jj = 0; jj < N / 2; jj++) {
                                const double2 t2 = EXA_LD2(off + v * NTS * SL + 2 * jj);
                                qw[2 * jj][v] = t2.x;
                                qw[2 * jj + 1][v] = t2.y;
                            }
                    } else {
#pragma unroll
                        for (int j = 0; j < N; j++)
#pragma unroll
                            for (int v = 0; v < NV; v++) qw[j][v] = EXA_LD(off + v * NTS * SL + j * ps);
                    }
                }
                EXA_STAMP(1);
                __syncthreads();                                 // (R) every read of Q is done
                EXA_STAMP(2);
                if (mine) {
                    double s[N][NV];
                    pencil_sums<D, N, PDE>(qw, ops_here<N>(ops_raw)->DEO, idx[D], s);
                    if constexpr (pde_has_source<PDE>::value) {                 // q_t + div F = S(q): the x pencils carry -S(q) of their nodes
                        if constexpr (D == 0) {
#pragma unroll
                            for (int i = 0; i < N; i++) {
                                double Sq[NV];
                                PDE::source(qw[i], Sq);
#pragma unroll
                                for (int v = 0; v < NV; v++) s[i][v] -= Sq[v];
                            }
                        }
                    }
                    {
                    if constexpr (WIDE) {
#pragma unroll
                        for (int v = 0; v < NV; v++)
#pragma unroll
                            for (int ii = 0; ii < N / 2; ii++)
                                EXA_ST2(off + D * ASZ + v * NTS * SL + 2 * ii, s[2 * ii][v], s[2 * ii + 1][v]);
                    } else {
#pragma unroll
                        for (int i = 0; i < N; i++)
#pragma unroll
                            for (int v = 0; v < NV; v++) EXA_ST(off + D * ASZ + v * NTS * SL + i * ps, s[i][v]);
                    }
                    }
                }
            }
        });
        if (it == 0 && wave >= DIM * W0) __syncthreads();        // (R) for the waves without a pencil in iteration 0
    } else {
        double s[N][NV];                                         // direction 0 keeps its sums over the barrier
        int zoff = 0;
        bool did_x = false;
        static_for<0, DIM>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            // iteration 0: the iterate is constant in time, so every time slab has the same sums -- only the
            // l = 0 pencils are computed (the time update below then uses the row sums of T).  They are
            // packed on W0 waves per direction at the front of the workgroup, i.e. on different SIMDs
            // (the first waves of the three groups would share one SIMD and serialise).
            constexpr int W0 = (CPB * NF + 63) / 64;
            const int wave = tid >> 6;
            const int k0 = (wave - D * W0) * 64 + (tid & 63);      // iteration-0 task of this lane for direction D
            const bool mine = it > 0 ? (grp == D && d_task >= 0) : (wave >= D * W0 && wave < (D + 1) * W0 && k0 < CPB * NF);
            if (mine EXA_ABL_COND_SKIP_D) {
                constexpr int ps = G::pstride(D);
                const int c = it > 0 ? d_task / NN : k0 / NF;
                const int r = it > 0 ? d_task - c * NN : k0 - c * NF;
                const int l = it > 0 ? r / NF : 0, t = it > 0 ? r - l * NF : r;
                if constexpr (D == 0) did_x = true;
                const int off = c * CS + l * SL + G::pbase(D, t);
                // even-odd form of the centro-antisymmetric D (DgOps::DEO): half the FMAs --
                // s_i -+ s_{N-1-i} from F_j +- F_{N-1-j}; odd N: the middle node and the middle row on top.
                // (tried in round 2, measured and left as opt-in macros: the whole operator in SGPRs first -- one batch, one
                // wait -- and every LDS load of the pencil issued before the first flux; both lose to the form below, whose
                // waits hide behind the other two waves of the SIMD)
                constexpr int H = N / 2;
                constexpr int NE = H * N + H + 1;
#ifndef EXA_A_SLOAD_D
                const EXA_AS4 double* Em = ops_here<N>(ops_raw)->DEO;
#else                // measured 2.5 % slower at N = 6 (profiles/r02_stage_a_variants.txt): 44 more live SGPRs, the waits were hidden
                double Em[NE];
                sload<NE>(ops_here<N>(ops_raw)->DEO, Em);
#endif
                // pencils along the contiguous axis (ps == 1, N even): 16-byte LDS accesses -- the pencil
                // bases are 3*t 16-byte units apart, a permutation mod 16 for every hardware lane group, so
                // ds_read_b128 / ds_write_b128 are conflict-free where 8-byte accesses are 2-way conflicted
                constexpr bool WIDE = (ps == 1) && (N % 2 == 0);
                double qw[N][NV];
                if constexpr (WIDE) {
#pragma unroll
                    for (int v = 0; v < NV; v++)
#pragma unroll
                        for (int jj = 0; jj < N / 2; jj++) {
                            const double2 t2 = EXA_LD2(off + v * NTS * SL + 2 * jj);
                            qw[2 * jj][v] = t2.x;
                            qw[2 * jj + 1][v] = t2.y;
                        }
                } else {
#ifdef EXA_A_LOADS_FIRST   // (every load of the pencil first: 0.5 % slower at N = 6, more live registers)
                    // in the order the even-odd form consumes them: node j, its mirror, next j
#pragma unroll
                    for (int j = 0; j < (N + 1) / 2; j++)
#pragma unroll
                        for (int v = 0; v < NV; v++) {
                            qw[j][v] = EXA_LD(off + v * NTS * SL + j * ps);
                            if (j != N - 1 - j) qw[N - 1 - j][v] = EXA_LD(off + v * NTS * SL + (N - 1 - j) * ps);
                        }
#endif
                }
                {
                    double P[H > 0 ? H : 1][NV], M[H > 0 ? H : 1][NV], mid[NV];
#pragma unroll
                    for (int i = 0; i < H; i++)
#pragma unroll
                        for (int v = 0; v < NV; v++) P[i][v] = M[i][v] = 0.0;
#pragma unroll
                    for (int v = 0; v < NV; v++) mid[v] = 0.0;
#pragma unroll
                    for (int j = 0; j < H; j++) {
                        const int jm = N - 1 - j;                // mirror node
                        double aa[nz(NA)], ab[nz(NA)], Fa[NV], Fb[NV];
#ifndef EXA_A_LOADS_FIRST
                        if constexpr (!WIDE) {
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                qw[j][v] = EXA_LD(off + v * NTS * SL + j * ps);
                                qw[jm][v] = EXA_LD(off + v * NTS * SL + jm * ps);
                            }
                        }
#endif
                        PDE::aux_fast(qw[j], aa);
                        PDE::template flux_scaled<D>(qw[j], aa, idx[D], Fa);
                        PDE::aux_fast(qw[jm], ab);
                        PDE::template flux_scaled<D>(qw[jm], ab, idx[D], Fb);
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
                                EXA_FMA(P[i][v], ea, Fa[v]);
                                EXA_FMA(M[i][v], eb, Fb[v]);
                            }
                        }
                        if constexpr (N % 2 == 1) {
                            const double em = Em[j * N + 2 * H];
#pragma unroll
                            for (int v = 0; v < NV; v++) EXA_FMA(mid[v], em, Fb[v]);
                        }
                    }
                    if constexpr (N % 2 == 1) {                  // middle node
                        double aa[nz(NA)], Fa[NV];
#ifndef EXA_A_LOADS_FIRST
                        if constexpr (!WIDE) {
#pragma unroll
                            for (int v = 0; v < NV; v++) qw[H][v] = EXA_LD(off + v * NTS * SL + H * ps);
                        }
#endif
                        PDE::aux_fast(qw[H], aa);
                        PDE::template flux_scaled<D>(qw[H], aa, idx[D], Fa);
#pragma unroll
                        for (int i = 0; i < H; i++) {
                            const double ec = Em[H * N + i];
#pragma unroll
                            for (int v = 0; v < NV; v++) EXA_FMA(P[i][v], ec, Fa[v]);
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
                }
                if constexpr (pde_has_source<PDE>::value) {                     // q_t + div F = S(q): the x pencils carry -S(q) of their nodes
                    if constexpr (D == 0) {
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            double Sq[NV];
                            PDE::source(qw[i], Sq);
#pragma unroll
                            for (int v = 0; v < NV; v++) s[i][v] -= Sq[v];
                        }
                    }
                }
                if constexpr (D > 0 && WIDE) {
#pragma unroll
                    for (int v = 0; v < NV; v++)
#pragma unroll
                        for (int ii = 0; ii < N / 2; ii++)
                            EXA_ST2(off + D * ASZ + v * NTS * SL + 2 * ii, s[2 * ii][v], s[2 * ii + 1][v]);
                } else
                if constexpr (D > 0) {
#pragma unroll
                    for (int i = 0; i < N; i++)
#pragma unroll
                        for (int v = 0; v < NV; v++) EXA_ST(off + D * ASZ + v * NTS * SL + i * ps, s[i][v]);
                } else {
                    zoff = off;
                }
            }
        });
        EXA_STAMP(1);
        __syncthreads();                                             // (1) every read of Q is done; S_y (A) and S_z (B) are complete
        EXA_STAMP(2);
        // ---- time update, one straight-line copy per direction group (wave-uniform branch: every wave of the workgroup
        // runs exactly one copy, each copy contains barrier (2) once).  Between barriers (1) and (2) the x group writes
        // its sums over Q (dead now) while everybody already fetches S_y + S_z of its own variables, so that after (2)
        // only S_x is left to load.
        static_for<0, DIM>([&](auto gc) {
            constexpr int GI = decltype(gc)::value;
            constexpr int CNT = tcnt(GI), V0 = tv0(GI);
            if (grp == GI) {
                if constexpr (GI == 0) {
                    if (did_x) {                                     // Q := S_x
                        constexpr int ps = G::pstride(0);
#pragma unroll
                        for (int i = 0; i < N; i++)
#pragma unroll
                            for (int v = 0; v < NV; v++) EXA_ST(zoff + v * NTS * SL + i * ps, s[i][v]);
                    }
                }
#ifndef EXA_ABL_SKIP_T
                if (it == 0) {                                       // the iterate was constant in time: level 0, row sums of T
                    double Syz0[CNT > 0 ? CNT : 1];
                    if (t_node) {
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++) {
                            const int o = toff + ((V0 + vv) * NTS + 0) * SL;
                            double x = EXA_LD(o + ASZ);
                            if constexpr (DIM == 3) x += EXA_LD(o + 2 * ASZ);
                            Syz0[vv] = x;
                        }
                    }
                    EXA_STAMP(10);
                    __syncthreads();                                 // (2) S_x is in Q
                    EXA_STAMP(11);
                    if (t_node && CNT > 0) {
                        double Ts[N];
                        sload<N>(step_here<N>(step_raw)->Tsdt, Ts);
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++) {
                            const int o = toff + ((V0 + vv) * NTS + 0) * SL;
                            const double x = EXA_LD(o) + Syz0[vv];
                            const double uv = um[vv];
#pragma unroll
                            for (int lp = 0; lp < N; lp++) EXA_ST(toff + ((V0 + vv) * NTS + lp) * SL, fma(Ts[lp], x, uv));
                        }
                    }
                } else {
                    double S[CNT > 0 ? CNT : 1][N];
                    if (t_node) {
                        double Sz[CNT > 0 ? CNT : 1][N];             // every load first, then the adds
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++)
#pragma unroll
                            for (int l = 0; l < N; l++) {
                                const int o = toff + ((V0 + vv) * NTS + l) * SL;
                                S[vv][l] = EXA_LD(o + ASZ);
                                if constexpr (DIM == 3) Sz[vv][l] = EXA_LD(o + 2 * ASZ);
                            }
                        if constexpr (DIM == 3) {
#pragma unroll
                            for (int vv = 0; vv < CNT; vv++)
#pragma unroll
                                for (int l = 0; l < N; l++) S[vv][l] += Sz[vv][l];
                        }
                    }
                    EXA_STAMP(10);
                    __syncthreads();                                 // (2) S_x is in Q
                    EXA_STAMP(11);
                    if (t_node && CNT > 0) {
                        double Sx[CNT > 0 ? CNT : 1][N];
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++)
#pragma unroll
                            for (int l = 0; l < N; l++) Sx[vv][l] = EXA_LD(toff + ((V0 + vv) * NTS + l) * SL);
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++)
#pragma unroll
                            for (int l = 0; l < N; l++) S[vv][l] += Sx[vv][l];
                        // -dt T, half of its rows at a time (N*N/2 doubles in SGPRs: the whole matrix next to the phase's
                        // other scalars would spill), each half serving every variable
                        constexpr int RH = (N + 1) / 2;
                        static_for<0, 2>([&](auto hc) {
                            constexpr int half = decltype(hc)::value;
                            constexpr int R0 = half * RH, RN = (half == 0) ? RH : N - RH;
                            double Tm[RN * N];
                            sload<RN * N>(step_here<N>(step_raw)->Tdt + R0 * N, Tm);
#pragma unroll
                            for (int vv = 0; vv < CNT; vv++) {
                                const double uv = um[vv];
#pragma unroll
                                for (int r = 0; r < RN; r++) {
                                    double acc = uv;
#pragma unroll
                                    for (int l = 0; l < N; l++) EXA_FMA(acc, Tm[r * N + l], S[vv][l]);
                                    EXA_ST(toff + ((V0 + vv) * NTS + R0 + r) * SL, acc);
                                }
                            }
                        });
                    }
                }
#else
                __syncthreads();
#endif
            }
        });
    }
    if constexpr (SA::MODE != 0) {
        EXA_STAMP(10);
        __syncthreads();                                             // (1) S_x (over Q), S_y (A), S_z (B) are complete
        EXA_STAMP(11);
        // ---- time update, one straight-line copy per direction group
        static_for<0, DIM>([&](auto gc) {
            constexpr int GI = decltype(gc)::value;
            constexpr int CNT = tcnt(GI), V0 = tv0(GI);
            if constexpr (CNT > 0) {
                if (grp == GI && t_node) {
                    if (it == 0) {                                   // the iterate was constant in time: level 0, row sums of T
                        double Ts[N];
                        sload<N>(step_here<N>(step_raw)->Tsdt, Ts);
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++) {
                            const int o = toff + ((V0 + vv) * NTS + 0) * SL;
                            double x = EXA_LD(o) + EXA_LD(o + ASZ);
                            if constexpr (DIM == 3) x += EXA_LD(o + 2 * ASZ);
                            const double uv = um[vv];
#pragma unroll
                            for (int lp = 0; lp < N; lp++) EXA_ST(toff + ((V0 + vv) * NTS + lp) * SL, fma(Ts[lp], x, uv));
                        }
                    } else {
                        double S[CNT][N], Sy[CNT][N], Sz[CNT][N];   // every load first, then the adds
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++)
#pragma unroll
                            for (int l = 0; l < N; l++) {
                                const int o = toff + ((V0 + vv) * NTS + l) * SL;
                                S[vv][l] = EXA_LD(o);
                                Sy[vv][l] = EXA_LD(o + ASZ);
                                if constexpr (DIM == 3) Sz[vv][l] = EXA_LD(o + 2 * ASZ);
                            }
#pragma unroll
                        for (int vv = 0; vv < CNT; vv++)
#pragma unroll
                            for (int l = 0; l < N; l++) {
                                S[vv][l] += Sy[vv][l];
                                if constexpr (DIM == 3) S[vv][l] += Sz[vv][l];
                            }
                        constexpr int RH = (N + 1) / 2;
                        static_for<0, 2>([&](auto hc) {
                            constexpr int half = decltype(hc)::value;
                            constexpr int R0 = half * RH, RN = (half == 0) ? RH : N - RH;
                            double Tm[RN * N];
                            sload<RN * N>(step_here<N>(step_raw)->Tdt + R0 * N, Tm);
#pragma unroll
                            for (int vv = 0; vv < CNT; vv++) {
                                const double uv = um[vv];
#pragma unroll
                                for (int r = 0; r < RN; r++) {
                                    double acc = uv;
#pragma unroll
                                    for (int l = 0; l < N; l++) EXA_FMA(acc, Tm[r * N + l], S[vv][l]);
                                    EXA_ST(toff + ((V0 + vv) * NTS + R0 + r) * SL, acc);
                                }
                            }
                        });
                    }
                }
            }
        });
    }
        EXA_STAMP(3);
        __syncthreads();                                             // (3) the new iterate is in Q
        EXA_STAMP(4);
    }

    // ---- time averages (A.3) per node, one direction per group (all waves busy; q is re-read by each group):
    //      Fbar_d -> A slab 1+d; group 0 also qbar -> A slab 0 and u -> B slab 0 (S_y, S_z are dead).
    //      One straight-line copy per group: the weights in one scalar batch, every LDS load of the node first.
    {
        const int c = bt / NN, n = bt - c * NN;
        const int off = c * CS + G::node_off(n);
        static_for<0, DIM>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            if (grp == D && bt < TD) {
                double qb[NV], Fb[NV];
                [[maybe_unused]] double Sb[NV];
                if constexpr (pde_has_source<PDE>::value) {
#pragma unroll
                    for (int v = 0; v < NV; v++) Sb[v] = 0.0;
                }
                // all variables of u at this node: for the single-stage case, and (group 0) for the B slab the final sum reads;
                // requested first, used last (an L2 hit: the cell was read at its start)
                double ur[NV];
                if (D == 0 || n_it <= 0) {
                    const long cell = cell_id[c];
#pragma unroll
                    for (int v = 0; v < NV; v++) ur[v] = cell >= 0 ? u_in[(cell * NN + n) * NV + v] : 1.0;
                }
                if (n_it > 0) {
                    double wm[N];
                    sload<N>(ops_here<N>(ops_raw)->w, wm);
                    double q[NV], qn[NV];                                // level l in use, level l + 1 in flight
#pragma unroll
                    for (int v = 0; v < NV; v++) qn[v] = EXA_LD(off + (v * NTS + 0) * SL);
#pragma unroll
                    for (int v = 0; v < NV; v++) qb[v] = Fb[v] = 0.0;
#pragma unroll
                    for (int l = 0; l < N; l++) {
                        double a[nz(NA)], F[NV];
#pragma unroll
                        for (int v = 0; v < NV; v++) q[v] = qn[v];
                        if (l + 1 < N) {
#pragma unroll
                            for (int v = 0; v < NV; v++) qn[v] = EXA_LD(off + (v * NTS + l + 1) * SL);
                        }
                        PDE::aux_fast(q, a);
                        if constexpr (D == 0) {
#pragma unroll
                            for (int v = 0; v < NV; v++) qb[v] += wm[l] * q[v];
                        }
                        if constexpr (pde_has_source<PDE>::value) {          // time average of the source
                            if constexpr (D == 0) {
                                double Sq[NV];
                                PDE::source(q, Sq);
#pragma unroll
                                for (int v = 0; v < NV; v++) Sb[v] += wm[l] * Sq[v];
                            }
                        }
                        PDE::template flux<D>(q, a, F);
#pragma unroll
                        for (int v = 0; v < NV; v++) Fb[v] += wm[l] * F[v];
                    }
                } else {
                    double a[nz(NA)];
                    PDE::aux_fast(ur, a);
#pragma unroll
                    for (int v = 0; v < NV; v++) qb[v] = ur[v];
                    PDE::template flux<D>(ur, a, Fb);
                    if constexpr (pde_has_source<PDE>::value) {
                        if constexpr (D == 0) PDE::source(ur, Sb);
                    }
                }
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    lds[off + ASZ + (v * NTS + 1 + D) * SL] = Fb[v];
                    if constexpr (D == 0) {
                        lds[off + ASZ + (v * NTS + 0) * SL] = qb[v];
                        // same (cell, node) as the update role of this lane; with a source: u + dt * (time-averaged source), what the
                        // final sum adds the volume terms to
                        if constexpr (pde_has_source<PDE>::value) lds[off + 2 * ASZ + (v * NTS + 0) * SL] = ur[v] + dt * Sb[v];
                        else lds[off + 2 * ASZ + (v * NTS + 0) * SL] = ur[v];
                    }
                }
            }
        });
    }
    __syncthreads();
    EXA_STAMP(7);

    // ---- volume integral + face extrapolation: group d takes the pencils of direction d, tasks (c, v, t), t fastest.
    //      The direction is a compile-time property of the copy (constant strides); Kxi in its even-odd form (it is
    //      centro-antisymmetric like D: half the FMAs, 22 instead of 36 scalars at N = 6), all scalars of the phase in two
    //      batches, the 2N LDS loads of the pencil first.
    static_for<0, DIM>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        if (grp == D) {
            constexpr int ps = G::pstride(D);
            constexpr int TDV = CPB * NV * NF;
            constexpr int NE = (N / 2) * N + N / 2 + 1;
            double KE[NE], iwm[N];
            sload<NE>(ops_here<N>(ops_raw)->KEO, KE);
            sload<N>(ops_here<N>(ops_raw)->iw, iwm);
            const double sc = dt * idx[D];
            for (int task = bt; task < TDV; task += GW) {
                const int c = task / (NV * NF);
                const int r = task - c * (NV * NF);
                const int v = r / NF, t = r - v * NF;
                const int off = c * CS + G::pbase(D, t) + v * NTS * SL;
                double qb[N], Fb[N], vol[N];
#pragma unroll
                for (int j = 0; j < N; j++) {
                    qb[j] = EXA_LD(off + ASZ + j * ps);
                    Fb[j] = EXA_LD(off + ASZ + (1 + D) * SL + j * ps);
                }
                eo_apply<N>(KE, Fb, vol);
#pragma unroll
                for (int i = 0; i < N; i++) lds[off + (1 + D) * SL + i * ps] = sc * iwm[i] * vol[i];
                double pl[N], pr[N];
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
                const long cell = cell_id[c];
                if (cell >= 0) {
                    double* tl = trace + (((long)D * 2 + 0) * ncells + cell) * (2 * NV * NF);
                    double* tr = trace + (((long)D * 2 + 1) * ncells + cell) * (2 * NV * NF);
                    tl[(0 * NV + v) * NF + t] = qL;
                    tl[(1 * NV + v) * NF + t] = FL;
                    tr[(0 * NV + v) * NF + t] = qR;
                    tr[(1 * NV + v) * NF + t] = FR;
                }
            }
        }
    });
    __syncthreads();
    EXA_STAMP(8);

    // ---- u* = u + sum_d vol_d, written AoS (coalesced)
    for (int task = tid; task < CPB * NN * NV; task += NT) {
        const int c = task / (NN * NV), e = task - c * (NN * NV);
        const int n = e / NV, v = e - n * NV;
        const int off = c * CS + G::node_off(n) + v * NTS * SL;
        double us = EXA_LD(off + 2 * ASZ);                       // u (B slab 0)
#pragma unroll
        for (int d = 0; d < DIM; d++) us += EXA_LD(off + (1 + d) * SL);
        const long cell = cell_id[c];
        if (cell >= 0) u_out[cell * (NN * NV) + e] = us;
    }
    EXA_STAMP(9);
    __syncthreads();                                             // LDS and the cell-id slot are reused by the next block
#pragma unroll
    for (int vv = 0; vv < NVA; vv++) um[vv] = unm[vv];
    }
    EXA_STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------
// Stage A, single-stage variant (n_picard = 0: qbar := u, Fbar := f(u); BASELINE configs[1])
// ------------------------------------------------------------------------------------------
// No space-time image: per cell only qbar and Fbar_d live in LDS ((1+DIM) node arrays per variable,
// the volume term overwrites Fbar_d pencil by pencil), so many cells share a workgroup and several
// workgroups share a CU -- this variant is HBM-bound (B_A = 16 m N^d + 32 d m N^(d-1) bytes per cell).
template <int DIM, int N, class PDE, int CPB, int NT>
__global__ void __launch_bounds__(NT)
dg_stage_a_single_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, double* __restrict__ trace,
                         long ncells, CellBox box, double dt, double idx0, double idx1, double idx2,
                         const void* __restrict__ ops_raw) {
    using G = Geo<DIM, N>;
    constexpr int NV = PDE::NV, NA = PDE::NAUX;
    constexpr int NN = G::NN, NF = G::NF, SL = G::SL;
    constexpr int CS = (1 + DIM) * NV * SL;                  // doubles per cell image
    constexpr int KMAX = (CPB * NN + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) double lds[CPB * CS];
    __shared__ long cell_id[CPB];
    const int tid = threadIdx.x;
    const long b0 = (long)blockIdx.x * CPB;
    const double idx[3] = {idx0, idx1, idx2};
    if (tid < CPB) cell_id[tid] = box.cell(b0 + tid);
    __syncthreads();

    double ur[KMAX][NV];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int task = tid + k * NT;
        if (task < CPB * NN) {
            const int c = task / NN, n = task - c * NN;
            const long cell = cell_id[c];
            const int off = c * CS + G::node_off(n);
            double a[nz(NA)], F[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) ur[k][v] = cell >= 0 ? u_in[(cell * NN + n) * NV + v] : 1.0;
            PDE::aux_fast(ur[k], a);
#pragma unroll
            for (int v = 0; v < NV; v++) lds[off + v * SL] = ur[k][v];
            static_for<0, DIM>([&](auto dc) {
                constexpr int D = decltype(dc)::value;
                PDE::template flux<D>(ur[k], a, F);
#pragma unroll
                for (int v = 0; v < NV; v++) lds[off + ((1 + D) * NV + v) * SL] = F[v];
            });
        }
    }
    __syncthreads();
    {
        const EXA_AS4 DgOps<N>* o = ops_here<N>(ops_raw);
        for (int task = tid; task < CPB * DIM * NV * NF; task += NT) {
            const int c = task / (DIM * NV * NF);
            int r = task - c * (DIM * NV * NF);
            const int d = r / (NV * NF);
            r -= d * (NV * NF);
            const int v = r / NF, t = r - v * NF;
            const int ps = G::pstride(d);
            const int off = c * CS + G::pbase(d, t);
            double qb[N], Fb[N];
#pragma unroll
            for (int j = 0; j < N; j++) {
                qb[j] = lds[off + v * SL + j * ps];
                Fb[j] = lds[off + ((1 + d) * NV + v) * SL + j * ps];
            }
            const double sc = dt * idx[d];
#pragma unroll
            for (int i = 0; i < N; i++) {
                double sv = 0.0;
#pragma unroll
                for (int j = 0; j < N; j++) sv += o->Kxi[i * N + j] * Fb[j];
                lds[off + ((1 + d) * NV + v) * SL + i * ps] = sc * o->iw[i] * sv;   // own pencil only
            }
            double qL = 0.0, qR = 0.0, FL = 0.0, FR = 0.0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                qL += o->phiL[j] * qb[j];
                qR += o->phiR[j] * qb[j];
                FL += o->phiL[j] * Fb[j];
                FR += o->phiR[j] * Fb[j];
            }
            const long cell = cell_id[c];
            if (cell >= 0) {
                double* tl = trace + (((long)d * 2 + 0) * ncells + cell) * (2 * NV * NF);
                double* tr = trace + (((long)d * 2 + 1) * ncells + cell) * (2 * NV * NF);
                tl[(0 * NV + v) * NF + t] = qL;
                tl[(1 * NV + v) * NF + t] = FL;
                tr[(0 * NV + v) * NF + t] = qR;
                tr[(1 * NV + v) * NF + t] = FR;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int task = tid + k * NT;
        if (task < CPB * NN) {
            const int c = task / NN, n = task - c * NN;
            const long cell = cell_id[c];
            const int off = c * CS + G::node_off(n);
            if (cell >= 0) {
                [[maybe_unused]] double Sq[NV];
                if constexpr (pde_has_source<PDE>::value) PDE::source(ur[k], Sq);     // single stage: + dt S(u)
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    double us = ur[k][v];
                    if constexpr (pde_has_source<PDE>::value) us += dt * Sq[v];
#pragma unroll
                    for (int d = 0; d < DIM; d++) us += lds[off + ((1 + d) * NV + v) * SL];
                    u_out[(cell * NN + n) * NV + v] = us;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stage B
// ------------------------------------------------------------------------------------------
// physical coordinates of face node y of the face (d, side) of cell cc: the normal coordinate is the face's, the others the nodes'
template <int DIM, int N> __device__ inline void face_node_coords(const PlainGeo& g, const long* cc, int d, int side, int y, double* x) {
    int tr[2] = {0, 0};                                                // the remaining axes in ascending order, their node indices
    if constexpr (DIM == 3) { tr[0] = y / N; tr[1] = y % N; }
    else tr[0] = y;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (a >= DIM) x[a] = 0.0;
        else if (a == d) x[a] = g.x0[a] + (double)(cc[a] + side) * g.h[a];
        else { x[a] = g.x0[a] + ((double)cc[a] + g.xi[tr[k]]) * g.h[a]; k++; }
    }
}

struct StageBArgs {
    long nc[3];            // local block
    long lo[3], nb[3];     // box origin and extent
    const double* ghost[6];
    int tiled;             // box slots enumerated tile by tile (every nb[d] a multiple of the tile edge), else lexicographically
    double* lam;           // CFL instantiations only: lam[0] = max(lam[0], largest eigenvalue of the corrected u over nodes and directions); zeroed by the caller
};

// r5: the CFL scan of the NEXT step inside the kernel that writes u (template flag CFL; the instantiation without it is the one every fixed-dt step
// launches, unchanged).  The corrector update is element-wise (lane = one variable of one node, coalesced), the eigenvalue needs a node's whole state:
// the new values pass through an LDS copy of the cell(s), one lane per node evaluates the eigenvalues, one integer atomic max per wave
// (non-negative doubles order like their bit patterns; a NaN is kept: nan_max).  Replaces a pass of its own over u (dg_maxeig_kernel: 2 % of a step of
// AderDgSolver.run at 128^3 cells, p = 5).
template <int DIM, class PDE, int CPB, int NN, int NT>
__device__ inline void stage_b_cfl_scan(const double* unode, long b0, long nbox, double* lam) {
    constexpr int NV = PDE::NV;
    const int tid = threadIdx.x;
    double m = 0.0;
    for (int nd = tid; nd < CPB * NN; nd += NT) {
        if (b0 + nd / NN >= nbox) continue;
        double q[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) q[v] = unode[nd * NV + v];
#pragma unroll
        for (int d = 0; d < DIM; d++) m = nan_max(m, PDE::maxeig(q, d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o, 64));
    // one atomic per wave would be millions on ONE address (96 against 15 ms per 128^3 launch): the maximum only grows, so a wave whose value does not
    // exceed what is there already (an L2-coherent read) has nothing to add -- after the first few workgroups almost every wave
    if ((tid & 63) == 0) {
        unsigned long long* lb = reinterpret_cast<unsigned long long*>(lam);
        const unsigned long long mine = (unsigned long long)__double_as_longlong(m);
        if (mine > __hip_atomic_load(lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(lb, mine);
    }
}
#ifndef EXA_STAGE_B_TILE_SHIFT
#define EXA_STAGE_B_TILE_SHIFT 3
#endif

__host__ __device__ constexpr int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// Faces are laid out on aligned lane segments of GS = pow2ceil(Nf) lanes, 64/GS faces per wave pass:
// the face-wide maximum eigenvalue (A.4) is a segmented butterfly of wavefront shuffles, no LDS, no
// barrier; only the Rusanov fluxes go through LDS to reach the node-major corrector update.
template <int DIM, int N, class PDE, int CPB, int NT, bool CFL = false>
__global__ void __launch_bounds__(NT)
dg_stage_b_kernel(double* __restrict__ u, const double* __restrict__ trace, StageBArgs A, long ncells, long nbox,
                  double dt, double idx0, double idx1, double idx2, DgOps<N> ops) {
    using G = Geo<DIM, N>;
    constexpr int NV = PDE::NV;
    constexpr int NN = G::NN, NF = G::NF;
    __shared__ double unode[CFL ? CPB * NN * NV : 1];
    constexpr int NFACE = 2 * DIM;
    constexpr int TS = 2 * NV * NF;                     // doubles per (cell, d, side) trace
    constexpr int GS = pow2ceil(NF);                    // lanes per face segment
    static_assert(GS <= 64, "a face must fit one wavefront");
    constexpr int FPB = NT / GS;                        // faces per workgroup pass
    constexpr int NFC = CPB * NFACE;                    // faces of this workgroup
    constexpr int KB = (NFC + FPB - 1) / FPB;
    __shared__ double fs[CPB * NFACE * NF * NV];
    __shared__ double cL[DIM][N], cR[DIM][N];

    const int tid = threadIdx.x;
    const long b0 = xcd_contiguous(blockIdx.x, gridDim.x) * CPB;
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
        if (DIM == 3 && A.tiled) {
            constexpr int TSH = EXA_STAGE_B_TILE_SHIFT, TM = (1 << TSH) - 1;
            const long t = b >> (3 * TSH);
            const int w = (int)(b & ((1 << (3 * TSH)) - 1));
            const long tz = A.nb[2] >> TSH, ty = A.nb[1] >> TSH;
            cz = ((t % tz) << TSH) + (w & TM);
            cy = (((t / tz) % ty) << TSH) + ((w >> TSH) & TM);
            cx = ((t / (tz * ty)) << TSH) + (w >> (2 * TSH));
        } else {
            if constexpr (DIM == 3) {
                cz = b % A.nb[2];
                b /= A.nb[2];
            }
            cy = b % A.nb[1];
            cx = b / A.nb[1];
        }
        cc[0] = A.lo[0] + cx;
        cc[1] = A.lo[1] + cy;
        cc[2] = (DIM == 3) ? A.lo[2] + cz : 0;
        return (cc[0] * A.nc[1] + cc[1]) * A.nc[2] + cc[2];
    };

    const int seg = tid / GS, y = tid - seg * GS;
#pragma unroll
    for (int k = 0; k < KB; k++) {
        const int face = seg + k * FPB;                 // (c, f) flattened
        const bool ok = face < NFC && y < NF;
        double qm[NV], qp[NV], Fm[NV], Fp[NV];
        double lam = 0.0;
        int d = 0;
        if (ok) {
            const int c = face / NFACE, f = face - c * NFACE;
            d = f >> 1;
            const int side = f & 1;
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
            if (side == 0) {   // minus = left neighbour's R trace, plus = own L trace
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
                qm[v] = pm[v * NF + y];
                qp[v] = pp[v * NF + y];
                Fm[v] = pm[(NV + v) * NF + y];
                Fp[v] = pp[(NV + v) * NF + y];
            }
#if EXA_STAGE_B_FAST_EIG
            lam = fmax(PDE::maxeig_fast(qm, d), PDE::maxeig_fast(qp, d));
#else
            lam = fmax(PDE::maxeig(qm, d), PDE::maxeig(qp, d));
#endif
        }
        // face-wide maximum: segmented butterfly over the GS lanes of the face (inactive lanes carry 0)
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) lam = fmax(lam, __shfl_xor(lam, o, GS));
        if (ok) {
#pragma unroll
            for (int v = 0; v < NV; v++)
                fs[(face * NF + y) * NV + v] = 0.5 * (Fm[v] + Fp[v]) - 0.5 * lam * (qp[v] - qm[v]);
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
            const int i = G::coord(n, d), yy = G::face_index(n, d);
            const double fR = fs[((c * NFACE + d * 2 + 1) * NF + yy) * NV + v];
            const double fL = fs[((c * NFACE + d * 2 + 0) * NF + yy) * NV + v];
            un -= cR[d][i] * fR - cL[d][i] * fL;
        }
        u[cell * (NN * NV) + e] = un;
        if constexpr (CFL) unode[task] = un;
    }
    if constexpr (CFL) {
        __syncthreads();
        stage_b_cfl_scan<DIM, PDE, CPB, NN, NT>(unode, b0, nbox, A.lam);
    }
}

// Dense variant for face sizes that are not a power of two (Nf = 9, 25, 36, 49): tasks (cell, face, node)
// packed densely over the lanes, face-wide maximum through a small LDS array (a 36-node face on a 64-lane
// segment would idle 44 % of the lanes of the trace loads).
template <int DIM, int N, class PDE, int CPB, int NT, bool CFL = false>
__global__ void __launch_bounds__(NT)
dg_stage_b_dense_kernel(double* __restrict__ u, const double* __restrict__ trace, StageBArgs A, long ncells, long nbox,
                  double dt, double idx0, double idx1, double idx2, DgOps<N> ops, PlainGeo geo) {
    using G = Geo<DIM, N>;
    constexpr int NV = PDE::NV;
    constexpr int NN = G::NN, NF = G::NF;
    __shared__ double unode[CFL ? CPB * NN * NV : 1];
    constexpr int NFACE = 2 * DIM;
    constexpr int TS = 2 * NV * NF;                     // doubles per (cell, d, side) trace
    constexpr int KB = (CPB * NFACE * NF + NT - 1) / NT;
    __shared__ double lam[CPB * NFACE * NF];
    __shared__ double fs[CPB * NFACE * NF * NV];
    __shared__ double cL[DIM][N], cR[DIM][N];

    const int tid = threadIdx.x;
    const long b0 = xcd_contiguous(blockIdx.x, gridDim.x) * CPB;
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
        if (DIM == 3 && A.tiled) {
            constexpr int TSH = EXA_STAGE_B_TILE_SHIFT, TM = (1 << TSH) - 1;
            const long t = b >> (3 * TSH);
            const int w = (int)(b & ((1 << (3 * TSH)) - 1));
            const long tz = A.nb[2] >> TSH, ty = A.nb[1] >> TSH;
            cz = ((t % tz) << TSH) + (w & TM);
            cy = (((t / tz) % ty) << TSH) + ((w >> TSH) & TM);
            cx = ((t / (tz * ty)) << TSH) + (w >> (2 * TSH));
        } else {
            if constexpr (DIM == 3) {
                cz = b % A.nb[2];
                b /= A.nb[2];
            }
            cy = b % A.nb[1];
            cx = b / A.nb[1];
        }
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
            if constexpr (pde_has_xt<PDE>::value) {
                double xf[3];
                face_node_coords<DIM, N>(geo, cc, d, face, y, xf);
                lam[task] = fmax(PDE::maxeig_xt(qm[k], xf, geo.t + 0.5 * dt, d), PDE::maxeig_xt(qp[k], xf, geo.t + 0.5 * dt, d));
            } else {
#if EXA_STAGE_B_FAST_EIG
                lam[task] = fmax(PDE::maxeig_fast(qm[k], d), PDE::maxeig_fast(qp[k], d));
#else
                lam[task] = fmax(PDE::maxeig(qm[k], d), PDE::maxeig(qp[k], d));
#endif
            }
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
            if constexpr (pde_has_ncp<PDE>::value) {
                // path-conservative jump term D = B_d((q- + q+)/2) (q+ - q-) (straight path, midpoint), half to either side: the cell left
                // of the face (this is its high face) takes F* + D/2, the cell right of it (its low face) F* - D/2
                const int c = task / (NFACE * NF), r = task - c * (NFACE * NF), f = r / NF, y = r - f * NF;
                const int d = f >> 1, face = f & 1;
                double qa[NV], dq[NV], Dj[NV], xf[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    qa[v] = 0.5 * (qm[k][v] + qp[k][v]);
                    dq[v] = qp[k][v] - qm[k][v];
                    Dj[v] = 0.0;
                }
                if constexpr (pde_has_xt<PDE>::value) {
                    long cc[3];
                    cell_of(c, cc);
                    face_node_coords<DIM, N>(geo, cc, d, face, y, xf);
                }
                fv_ncp<PDE>(qa, dq, xf, geo.t + 0.5 * dt, d, Dj);
#pragma unroll
                for (int v = 0; v < NV; v++) fs[task * NV + v] += (face == 1 ? 0.5 : -0.5) * Dj[v];
            }
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
        if constexpr (CFL) unode[task] = un;
    }
    if constexpr (CFL) {
        __syncthreads();
        stage_b_cfl_scan<DIM, PDE, CPB, NN, NT>(unode, b0, nbox, A.lam);
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
        for (int d = 0; d < DIM; d++) m = nan_max(m, PDE::maxeig(q, d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o, 64));
    __shared__ double wm[16];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = nan_max(m, wm[w]);
        // non-negative doubles order like their bit patterns -> integer atomic max
        atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(m));
    }
}

}  // namespace exa
