// Host-side glue between the C-ABI (capi.cpp) and the per-(dim, pde) kernel
// instantiation units (dg_inst.hip, fv_rusanov.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "exa_dg_common.hpp"

namespace exa {

// reference-element tables for any N <= MAXN (host copy held by the plan)
struct DgOpsHost {
    int N;
    double xi[MAXN], w[MAXN], D[MAXN * MAXN], Kxi[MAXN * MAXN], phiL[MAXN], phiR[MAXN], iK1[MAXN * MAXN],
        K1[MAXN * MAXN];
    void* dev;     // DgOps<N> image in HBM (read by the kernels through the constant address space)
    void* lim;     // limiter tables in HBM: P[Ns][N] then R[N][Ns] (null until first use)
    void* scratch; // per-workgroup slabs of the level-streamed stage A (N whose cell image exceeds LDS), else null
    int stage_a_reserve;   // workgroups the persistent stage-A grids stay below the resident count (CUs left to the exchange's kernels)
    int stage_a_variant;   // 0: the build's default; 1: LDS-resident space-time image; 2: register-resident iterate (where built: 3-D, N = 6)
    double origin[3];      // physical coordinates of the block's origin and the time at the start of the step: what term sets with
    double time;           // position- / time-dependent terms (HAS_XT) are evaluated at (exa_dg_plan_set_origin_time; default 0)
};
// dg_operators_host.cpp: Gauss-Legendre nodes by Newton, barycentric derivative
// matrix, K1 inverse by Gauss-Jordan, all in long double.
int build_dg_operators(int N, DgOpsHost* out);
int build_limiter_operators(const DgOpsHost* o, int Ns, double* P, double* R);     // P[Ns][N], R[N][Ns]
// limiter.hip
struct LimGhosts { const double* layer[6]; };   // [d*2+side]: neighbour block's subcell layer at that block face, or null
int limiter_project(int dim, int N, int Ns, int nv, const long* nc, const double* u, const long* cells, long n, double* patch,
                    const double* Pdev, const LimGhosts* ghosts, hipStream_t s);
int limiter_face_layers(int dim, int N, int Ns, int nv, const long* nc, const double* u, int a, int side, const double* need,
                        double* out, const double* Pdev, hipStream_t s);
int limiter_reconstruct(int dim, int N, int Ns, int nv, const double* patch, const long* cells, long n, double* u,
                        const double* Rdev, hipStream_t s);

struct StageBBox {
    long nc[3], lo[3], nb[3];
    const double* ghost[6];
    double* lam = nullptr;      // not null: the CFL scan of the corrected u rides in the launch (exa_dg_riemann_corrector_cfl); the caller zeroes it
};

struct DgLaunchTable {
    int nv;        // variables the PDE evolves
    int max_n;     // largest N instantiated
    int (*stage_a)(int N, const double* u_in, double* u_out, double* trace, long ncells, const CellBox* box, double dt,
                   const double* idx, int n_it, const DgOpsHost* ops, hipStream_t s);
    int (*stage_b)(int N, double* u, const double* trace, const StageBBox* box, long ncells, double dt,
                   const double* idx, const DgOpsHost* ops, hipStream_t s);
    int (*maxeig)(const double* u, long nnodes, double* out, hipStream_t s);
    size_t (*ops_image)(int N, const DgOpsHost* h, void* dst);   // dst == nullptr: size only
    size_t (*scratch_bytes)(int N);                               // 0: the LDS kernel serves this N
    const char* (*stage_a_name)(int N, int n_it, int variant);    // the kernel stage_a launches for these settings (profiles, bench line)
    // the step as ONE kernel: Riemann + corrector of the previous step (traces trace_in, time step dt_prev) in front of the predictor of
    // this one (traces -> trace_out, a second array); u holds u* before and after.  u_plain: where the corrected u goes too, or null
    int (*stage_ba)(int N, double* u, const double* trace_in, double* trace_out, long ncells, const CellBox* box, const double* const* ghost,
                    double dt_prev, double dt, const double* idx, int n_it, const DgOpsHost* ops, double* u_plain, hipStream_t s);
    int (*has_stage_ba)(int N, int n_it, int variant);
    // fused single-stage periodic step u_in -> u_out (2-D, n_picard = 0; exa_dg_fused.hpp), or nullptr
    int (*fused_single)(int N, const double* u_in, double* u_out, const long* nc, double dt, const double* idx,
                        const DgOpsHost* ops, hipStream_t s);
};
// returns nullptr when (dim, pde) is not built
const DgLaunchTable* dg_launch_table(int dim, int pde);

// fv_rusanov.hip
// slot: nullptr, or one entry per patch (< 0: patch not in use, left untouched)
// out != nullptr: out of place (QOut halo-less, [patch][P^dim][n_real + n_aux]); centre: [patch][dim] cell centres or nullptr; t: time
// grid != nullptr: the GRID step -- the patches are the cells of a Cartesian grid g[0] x g[1] (x g[2]) (row-major), halo states are taken from the
// face neighbours' interiors (or bstate[(axis * 2 + side) * V ..] on a domain face; null = periodic), results go to grid->out (layout of Q)
struct FvGridArgs {
    double* out;
    const double* bstate;
    int g[3];
    double* lam;          // optional: receives the largest eigenvalue of the NEW interior states (zeroed by fv_launch on the stream)
};
int fv_launch(int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, int pde, double* Q, double dt,
              double h, const long* slot, hipStream_t s, double* out = nullptr, const double* centre = nullptr, double t = 0.0,
              const FvGridArgs* grid = nullptr);
// max over the INTERIOR volumes of all patches and the directions of the largest eigenvalue -> lam[0] (device); centre / t / h as above
int fv_maxeig_launch(int dim, int P, int H, int n_real, int n_aux, long n_patches, int pde, const double* Q, double* lam, hipStream_t s,
                     const double* centre, double t, double h);
// X: [n][3] positions (term sets that depend on position / time) or null; t: time
int pde_eval_launch(int pde, int normal, long n, int stride, const double* Q, double* F, double* lam, hipStream_t s, const double* X = nullptr,
                    double t = 0.0);

// user PDE term sets registered at run time (capi.cpp: exa_register_pde), pde ids >= 100
int user_fv_launch(int pde, int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, double* Q, double dt,
                   double h, const long* slot, hipStream_t s, double* out, const double* centre, double t, const FvGridArgs* grid);
int user_fv_maxeig(int pde, int dim, int P, int H, int n_real, int n_aux, long n_patches, const double* Q, double* lam, hipStream_t s,
                   const double* centre, double t, double h);
int user_pde_eval(int pde, int normal, long n, int stride, const double* Q, double* F, double* lam, hipStream_t s, const double* X, double t);

void set_error(const char* fmt, ...);

}  // namespace exa
