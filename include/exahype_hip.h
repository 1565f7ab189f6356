/*
 * exahype_hip.h -- C-ABI of libexahype_hip.so, the MI355X (gfx950) drop-in for
 * the cell-local kernel stack of xdslproject/ExaHyPE.
 *
 * Conventions
 *  - every entry point returns 0 on success, a negative EXA_ERR_* otherwise;
 *    exa_last_error() gives the thread-local message.  Nothing throws across
 *    the boundary.
 *  - plain pointers and sizes only.  `*_dev` pointers are device (HBM)
 *    addresses owned by the caller; `*_host` pointers are host memory owned by
 *    the caller.  `stream` is a hipStream_t passed as void* (NULL = default).
 *  - all arithmetic is IEEE fp64.  Arrays use the reference layout: AoS,
 *    row-major, variable fastest (reference `exahype/printers/CPPPrinter.py:247-261`).
 *      FV : Q[patch][i][j]([k])[var], halo included, var < n_real + n_aux
 *      DG : u[cell][node][var], cell = (cx*ncy+cy)*ncz+cz, node = (i*N+j)*N+k
 *    axis 0 is the reference's index `i` (normal == 0).
 *  - a plan is bound to one device; distinct plans are independent; one plan is
 *    not thread-safe.
 *
 * Reference interfaces replaced
 *  - `void time_step(double* Q, double dt);`                  Unit test/test.h:3
 *        -> exa_fv_time_step_host / exa_fv_time_step_device (mode EXA_FV_FAITHFUL)
 *  - `void Flux(const double*, int normal, double* F);`        Unit test/Functions.h:2
 *    `double maxEigenvalue(const double*, int normal);`        Unit test/Functions.h:3
 *    `double max(double*, double*);`                           Unit test/Functions.h:4
 *        -> built-in device PDE terms selected by `pde` (EXA_PDE_*), exercised
 *           point-wise through exa_pde_eval_device
 *  - the generated loop nests `Unit test/test.cpp:11-104` (what
 *    exahype/printers/CPPPrinter.py:84-90 emits per KernelBuilder statement)
 *        -> the fused FV Rusanov patch kernel behind exa_fv_time_step_*
 *  - ADER-DG stages (space-time predictor, volume integral, face extrapolation,
 *    Rusanov Riemann solve, surface corrector): NOT in the reference
 *    (SURVEY.md F2); the north-star adds them.  -> exa_dg_*
 */
#ifndef EXAHYPE_HIP_H
#define EXAHYPE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXA_OK 0
#define EXA_ERR_INVALID (-1)     /* bad argument / unsupported configuration */
#define EXA_ERR_HIP (-2)         /* a HIP runtime call failed */
#define EXA_ERR_NO_DEVICE (-3)   /* no usable GPU */
#define EXA_ERR_ALLOC (-4)

/* point-wise PDE term sets (device twins of Unit test/Functions.cpp) */
#define EXA_PDE_EULER_REF2D 0    /* Functions.cpp:9-62 as compiled by the reference (2-D branch: reads Q[0..3], writes F[0..3]) */
#define EXA_PDE_EULER 1          /* same arithmetic, (rho, m0, m1, m2, E), gamma = 1.4 */
#define EXA_PDE_ADVECTION 2      /* linear advection of every variable, a = (1, 0.5, -0.75) */

/* FV Rusanov patch-update modes */
#define EXA_FV_FAITHFUL 0        /* Unit test/test.cpp:11-104 statement for statement (zero-initialised temporaries) */
#define EXA_FV_RUSANOV 1         /* corrected Rusanov: dt/h, all n_real variables, dissipative sign */

typedef struct exa_fv_plan exa_fv_plan;
typedef struct exa_dg_plan exa_dg_plan;

/* ---- library ---------------------------------------------------------------- */
int exa_version(void);
const char* exa_last_error(void);
int exa_device_count(int* count);

/* ---- user PDE term sets ------------------------------------------------------- */
/* The reference resolves `Flux` / `maxEigenvalue` at link time to user C++ (Unit test/Functions.h:2-3).
 * Here a user term set is a side library generated from SymPy expressions and compiled with hipcc
 * (exahype_amd/pde_codegen.py); registering it yields a pde id >= 100 usable wherever EXA_PDE_* is. */
int exa_register_pde(const char* library_path, int* pde_id);
/* What a registered term set carries (0 for the built-in ones): EXA_PDE_FLAG_XT -- its terms depend on position / time (they see the
 * coordinates the kernels hand them; exa_pde_eval_device and exa_dg_max_eigenvalue, which have none, evaluate them at x = 0, t = 0:
 * use exa_pde_eval_device_at for the CFL scan of such a term set, as exahype_amd/solvers.py does);
 * EXA_PDE_FLAG_NCP -- it carries a non-conservative product. */
#define EXA_PDE_FLAG_XT 1
#define EXA_PDE_FLAG_NCP 2
int exa_pde_flags(int pde);

/* ---- point-wise PDE terms (Functions.h:2-3) ---------------------------------- */
/* For n states Q_dev[n][stride] and a normal: F_dev[n][stride] (first n_flux
 * entries written) and lambda_dev[n].  Either output may be NULL. */
int exa_pde_eval_device(int pde, int normal, long n, int stride, const double* Q_dev, double* F_dev,
                        double* lambda_dev, void* stream);
/* The same with the positions of the states, x_dev [n][3] (NULL: the origin), and the time: for term sets whose terms depend on them
 * (EXA_PDE_FLAG_XT) -- the CFL scan of a host driver hands the volume centres / node coordinates here. */
int exa_pde_eval_device_at(int pde, int normal, long n, int stride, const double* Q_dev, const double* x_dev, double t, double* F_dev,
                           double* lambda_dev, void* stream);

/* ---- Finite-Volume Rusanov patch update (test.h:3) -------------------------- */
int exa_fv_plan_create(int device, int mode, int dim, int patch_size, int halo_size, int n_real, int n_aux,
                       long n_patches, int pde, exa_fv_plan** plan);
int exa_fv_plan_destroy(exa_fv_plan* plan);
/* doubles in one Q array: n_patches * (patch_size + 2*halo_size)^dim * (n_real + n_aux) */
long exa_fv_q_count(const exa_fv_plan* plan);
/* exact analogue of `time_step(Q, dt)`: host AoS array updated in place (interior
 * volumes only); the library stages it through HBM.  `h` (volume size) is used
 * by EXA_FV_RUSANOV only. */
int exa_fv_time_step_host(exa_fv_plan* plan, double* Q_host, double dt, double h);
/* same on a device-resident array (the hot path: no PCIe in the call) */
int exa_fv_time_step_device(exa_fv_plan* plan, double* Q_dev, double dt, double h, void* stream);
/* same over a patch array of which only some entries are in use, their number known on the device only (the
 * limiter's troubled cells): the launch covers the plan's n_patches (the array's capacity) and skips every
 * patch p with slot_dev[p] < 0.  slot_dev == NULL: all patches (== exa_fv_time_step_device). */
int exa_fv_time_step_device_masked(exa_fv_plan* plan, double* Q_dev, const long* slot_dev, double dt, double h,
                                   void* stream);
/* ... with the patch centres ([n_patches][dim], entries of unused patches ignored) and the time, for term sets whose terms depend on position /
 * time (the limiter's troubled cells of such a system: the centre of a patch is the centre of its DG cell) */
int exa_fv_time_step_device_masked_at(exa_fv_plan* plan, double* Q_dev, const long* slot_dev, const double* centre_dev, double t, double dt,
                                      double h, void* stream);

/* The `exahype2::CellData` flavour of the kernel (`examples/kernel-generator.py:6-45`; `exahype/KernelBuilder.py:217-218`: the output item is
 * indexed without the halo; `Unit test/correctness_test.cpp:142`: CellData(QIn, cellCentre, cellSize, t, dt, QOut)): QIn_dev
 * [n_patches][(P+2H)^dim][n_real+n_aux] is only read, QOut_dev [n_patches][P^dim][n_real+n_aux] (exa_fv_qout_count doubles) receives
 * the updated evolved variables and a copy of the auxiliary ones.  centre_dev: [n_patches][dim] cell centres (NULL: all at the origin)
 * and t reach PDE term sets that depend on position and time (flux / maxEigenvalue / sourceTerm(Q, x, h, t, dt, ...) of
 * `Unit test/correctness_test.cpp:16-41`; generated from SymPy expressions by exahype_amd.pde_codegen); the volume centres follow
 * exahype2::fv::getVolumeCentre: centre - P h / 2 + (index + 1/2) h. */
int exa_fv_time_step_device_oop(exa_fv_plan* plan, const double* QIn_dev, double* QOut_dev, const double* centre_dev, double t,
                                double dt, double h, void* stream);
/* The in-place update WITH the patch centres and the time (same meaning as for exa_fv_time_step_device_oop): what a host driver that keeps
 * Q with halo resident (halo fill -> update -> ...) calls for term sets whose terms depend on position / time.  exa_fv_time_step_device
 * is this call with every patch centred at the origin and t = 0. */
int exa_fv_time_step_device_at(exa_fv_plan* plan, double* Q_dev, const double* centre_dev, double t, double dt, double h, void* stream);
long exa_fv_qout_count(const exa_fv_plan* plan);

/* The patch update WITH its halo fill, on HALO-LESS arrays: the plan's patches are the cells of a Cartesian grid (grid[dim] patches per axis,
 * patch index row-major) and Q_dev / QNext_dev are [n_patches][P^dim][n_real+n_aux] (exa_fv_qout_count doubles each, the layout of the
 * CellData flavour's QOut).  What the reference leaves to the caller before `time_step` -- Peano keeps patches without halo and its enclave
 * task assembles QIn with halo from the neighbours' boundary layers (`exahype/printers/CPPPrinter.py:346`; the patch loop of
 * `Unit test/correctness_test.cpp:118-174`) -- happens inside the launch: the kernel builds the patch with halo on chip (LDS), taking the states
 * beyond a patch face from the face neighbour's interior layers in Q_dev (periodic wrap; boundary_dev != NULL: on a domain face the prescribed
 * state boundary_dev[(axis * 2 + side) * (n_real + n_aux) ..]).  Q_dev is only read; QNext_dev (a different array) receives the new states --
 * a driver swaps the two per step.  No halo pass, no halo bytes in HBM.  centre_dev, t as exa_fv_time_step_device_at.
 * lambda_next_dev (device, 1 double, or NULL): receives the largest eigenvalue of the NEW states over the directions (evaluated at t + dt) --
 * the CFL scan of the NEXT step, computed on the values the kernel holds in registers instead of by a pass of its own. */
int exa_fv_grid_step_device(exa_fv_plan* plan, const double* Q_dev, double* QNext_dev, const long* grid, const double* boundary_dev,
                            const double* centre_dev, double t, double dt, double h, double* lambda_next_dev, void* stream);
/* CFL scan of a patch array: max over the INTERIOR volumes of all patches and over the directions of the largest eigenvalue -> lambda_dev[0]
 * (device memory; one reduction launch, nothing returns to the host).  halo_less != 0: Q_dev is a halo-less array (every volume counts).
 * centre_dev / t / h: for term sets whose terms depend on position / time. */
int exa_fv_max_eigenvalue(exa_fv_plan* plan, const double* Q_dev, int halo_less, const double* centre_dev, double t, double h, double* lambda_dev,
                          void* stream);

/* ---- ADER-DG cell kernels ---------------------------------------------------- */
/* N = order + 1 nodes per axis; n_vars must equal the PDE's variable count (5 for
 * both Euler sets, any 1..8 for advection); n_picard < 0 selects N iterations,
 * 0 the single-stage variant (qbar := u, Fbar := f(u)); ncells[dim] is the local
 * Cartesian block of cells. */
int exa_dg_plan_create(int device, int dim, int N, int n_vars, int pde, int n_picard, const long* ncells,
                       exa_dg_plan** plan);
int exa_dg_plan_destroy(exa_dg_plan* plan);
/* Which stage-A kernel the plan launches where more than one is built for its (dim, N) -- today 3-D, N = 6:
 * EXA_STAGE_A_AUTO the library's choice (the faster one as measured on MI355X; the environment variable
 * EXA_STAGE_A=lds|reg, read when the plan is created, overrides it), EXA_STAGE_A_LDS the LDS-resident space-time
 * image (one workgroup per CU), EXA_STAGE_A_REG the register-resident iterate (two workgroups per CU).  Same scheme,
 * results equal to rounding.  Other (dim, N) ignore the setting. */
#define EXA_STAGE_A_AUTO 0
#define EXA_STAGE_A_LDS 1
#define EXA_STAGE_A_REG 2
int exa_dg_plan_set_stage_a(exa_dg_plan* plan, int variant);
/* Stage A of the larger orders runs as a persistent grid that fills every CU (one or two workgroups per CU, all of its LDS or
 * registers): kernels on OTHER streams -- the face packing and the RCCL transport of a halo exchange that is meant to overlap the
 * interior cells -- then only get CUs as those workgroups retire.  `workgroups` > 0 keeps the persistent grids that many workgroups
 * below the resident count, i.e. leaves that many CUs free (0, the default: fill the chip). */
int exa_dg_plan_set_stage_a_reserve(exa_dg_plan* plan, int workgroups);
/* name of the kernel exa_dg_predictor_volume launches for this plan, as a profiler prints it without the argument list
 * (static storage, valid until the next call); for bench lines and profile summaries */
const char* exa_dg_stage_a_kernel(const exa_dg_plan* plan);
long exa_dg_dof_count(const exa_dg_plan* plan);    /* doubles in u:      ncells * N^dim * n_vars */
long exa_dg_trace_count(const exa_dg_plan* plan);  /* doubles in traces: dim*2*ncells*2*n_vars*N^(dim-1) */
long exa_dg_face_count(const exa_dg_plan* plan, int d); /* doubles in one ghost/pack buffer for direction d */
/* host copies of the reference-element tables the plan uses (each may be NULL):
 * xi[N] w[N] D[N*N] Kxi[N*N] phiL[N] phiR[N] iK1[N*N] */
int exa_dg_operators(const exa_dg_plan* plan, double* xi, double* w, double* D, double* Kxi, double* phiL,
                     double* phiR, double* iK1);
/* algorithmic work of one stage-A launch (SURVEY.md 8(d) formulas), for rooflines */
int exa_dg_work(const exa_dg_plan* plan, double* flop_stage_a, double* flop_stage_b, double* bytes_stage_a,
                double* bytes_stage_b);

/* Stage A, all local cells: space-time predictor (Picard), time averages, volume
 * integral, face extrapolation.  u_dev is updated IN PLACE to u* ; trace_dev
 * receives trace[((d*2+side)*ncells + cell)*(2*n_vars*Nf) + (field*n_vars+v)*Nf + y]
 * (field 0 = time-averaged state, 1 = time-averaged normal flux; side 0 = xi=0). */
int exa_dg_predictor_volume(exa_dg_plan* plan, double* u_dev, double* trace_dev, double dt, const double* dx,
                            void* stream);
/* Stage A restricted to the cell box [lo, hi) of the local block (NULL = whole
 * block): lets a multi-GPU driver run the block's boundary shell first, start the
 * face-trace exchange, and overlap it with the interior cells. */
int exa_dg_predictor_volume_box(exa_dg_plan* plan, double* u_dev, double* trace_dev, const long* lo, const long* hi,
                                double dt, const double* dx, void* stream);
/* Stage B on the cell box [lo, hi) of the local block: Rusanov flux on the 2*dim
 * faces of every cell and the surface corrector; u_dev (holding u*) is updated
 * in place.  ghost_dev[d*2+0] / [d*2+1] are the neighbour block's traces across
 * the low / high block face in direction d, layout [transverse cell][2*n_vars*Nf]
 * (what exa_dg_pack_face produces on the neighbour); a NULL entry means periodic
 * wrap inside the block. */
int exa_dg_riemann_corrector(exa_dg_plan* plan, double* u_dev, const double* trace_dev,
                             const double* const* ghost_dev, const long* lo, const long* hi, double dt,
                             const double* dx, void* stream);
/* The same launch with the CFL scan of the NEXT step riding in it: *lambda_dev (one double, device) receives the maximum of
 * maxEigenvalue (`Unit test/Functions.h:3`) over the nodes of the box's corrected u and the directions -- what exa_dg_max_eigenvalue
 * would return for those cells, without a pass of its own over u.  lambda_dev is set to zero on the stream first.  Term sets whose
 * eigenvalue sees position / time (EXA_PDE_FLAG_XT) are refused: their scan needs the node coordinates (exa_pde_eval_device_at). */
int exa_dg_riemann_corrector_cfl(exa_dg_plan* plan, double* u_dev, const double* trace_dev,
                                 const double* const* ghost_dev, const long* lo, const long* hi, double dt,
                                 const double* dx, double* lambda_dev, void* stream);
/* Where and when: physical coordinates of the local block's origin (dim entries, NULL = 0) and the time at the start of the next
 * step.  Only term sets whose terms depend on position / time see them (pde_codegen.SympyPDE with flux(q, x, t, d) ...: the hooks
 * `Unit test/correctness_test.cpp:16-41` declares with (Q, x, h, t, dt)): stage A evaluates them at the nodes x = origin + (cell + xi_i) dx
 * and the level times t + xi_l dt, stage B at the face nodes and t + dt / 2.  Call it before every step of such a term set. */
int exa_dg_plan_set_origin_time(exa_dg_plan* plan, const double* origin, double t);
/* The step as ONE kernel ("fused" in the north star's sense; no counterpart in the reference): on the cell box
 * [lo, hi) every cell first finishes the PREVIOUS step -- Rusanov flux on its faces from the traces that step left in
 * trace_in_dev (ghost_dev as for exa_dg_riemann_corrector) and the surface corrector with dt_prev on u_dev (holding u*) --
 * and then runs stage A with dt on the result: u_dev holds the new u* afterwards, the new traces go to trace_out_dev, a
 * SECOND array (the neighbours still read trace_in_dev).  u_plain_dev, if not NULL, receives the corrected u of the previous
 * step (a snapshot).  A run of n steps is  predictor_volume, (n - 1) x corrector_predictor, riemann_corrector.
 * exa_dg_has_corrector_predictor: 1 where the plan's settings have this kernel (3-D, N = 6, register-resident stage A,
 * n_picard >= 1), else 0 -- exa_dg_corrector_predictor then returns EXA_ERR_INVALID. */
int exa_dg_has_corrector_predictor(const exa_dg_plan* plan);
int exa_dg_corrector_predictor(exa_dg_plan* plan, double* u_dev, const double* trace_in_dev, double* trace_out_dev,
                               const double* const* ghost_dev, const long* lo, const long* hi, double dt_prev, double dt,
                               const double* dx, double* u_plain_dev, void* stream);
/* Copy the outward traces of the block's boundary layer in direction d into a
 * contiguous buffer: side 0 -> L traces of the cells with c_d == 0 (to be sent
 * to the low neighbour, which uses it as ghost_dev[d*2+1]); side 1 -> R traces
 * of the cells with c_d == ncells[d]-1 (-> the high neighbour's ghost_dev[d*2+0]). */
int exa_dg_pack_face(exa_dg_plan* plan, const double* trace_dev, int d, int side, double* buf_dev, void* stream);
/* FV subcell limiter glue (BASELINE configs[4]): N_s = 2N-1 subcells per axis.
 * exa_lim_operators: host copies of the projection P[N_s][N] and the mean-preserving least-squares
 * reconstruction R[N][N_s].  exa_dg_project_patches: for the n cells listed in cells_dev build FV patches
 * patch_dev[n][(N_s+2)^dim][n_vars] (patch_size N_s, halo 1: interior = projected cell, face halos = adjacent
 * subcell layer of the projected face neighbours, periodic in the block) ready for exa_fv_time_step_device;
 * exa_dg_reconstruct_patches maps the patch interiors back onto the DG nodes of those cells.
 * cells_dev[i] < 0 marks an empty slot of a capacity-sized list (patch i is then neither written nor read): the list
 * can be compacted on the device without the host ever learning how many cells are troubled. */
int exa_lim_operators(const exa_dg_plan* plan, double* P, double* R);
long exa_lim_patch_count(const exa_dg_plan* plan);
int exa_dg_project_patches(exa_dg_plan* plan, const double* u_dev, const long* cells_dev, long n, double* patch_dev,
                           void* stream);
int exa_dg_reconstruct_patches(exa_dg_plan* plan, const double* patch_dev, const long* cells_dev, long n, double* u_dev,
                               void* stream);
/* Sharded grids ("a second exchange of one subcell halo layer for troubled boundary cells").
 * exa_lim_face_layers: for the cells of the block's boundary layer at face (d, side) write the projected subcell
 * layer adjacent to that face, out_dev[transverse cell][N_s^(dim-1)][n_vars] (exa_lim_face_layer_count(plan, d)
 * doubles); need_dev[transverse cell] != 0 selects the cells whose neighbour across the face is troubled (NULL: all;
 * skipped entries are left untouched).  The neighbour passes the received buffer as ghost_layers_dev[d*2+side'] of
 * its own face (side' = 1-side) to exa_dg_project_patches_ghost; NULL entries keep the periodic wrap in the block. */
long exa_lim_face_layer_count(const exa_dg_plan* plan, int d);
int exa_lim_face_layers(exa_dg_plan* plan, const double* u_dev, int d, int side, const double* need_dev, double* out_dev,
                        void* stream);
int exa_dg_project_patches_ghost(exa_dg_plan* plan, const double* u_dev, const long* cells_dev, long n, double* patch_dev,
                                 const double* const* ghost_layers_dev, void* stream);
/* max over all cells/nodes/directions of maxEigenvalue (for a CFL time step);
 * result is written to *lambda_dev (one double, device). */
int exa_dg_max_eigenvalue(exa_dg_plan* plan, const double* u_dev, double* lambda_dev, void* stream);
/* Single-stage scheme (n_picard = 0) in 2-D, alternative to predictor_volume + riemann_corrector on a periodic single
 * block: one fused launch per step with the traces kept on chip (exa_dg_fused.hpp); u_in_dev != u_out_dev (ping-pong).
 * Moves ~1.9 KB instead of 6.4 KB per cell and step through HBM at p = 3: 0.22 ms against 0.28 ms on 512 x 512 cells.
 * exa_dg_has_fused_step: 1 if the plan has this path. */
int exa_dg_has_fused_step(const exa_dg_plan* plan);
int exa_dg_step_fused(exa_dg_plan* plan, const double* u_in_dev, double* u_out_dev, double dt, const double* dx, void* stream);
/* convenience: n_steps full steps on a periodic single block (stage A + stage B) */
int exa_dg_step_periodic(exa_dg_plan* plan, double* u_dev, double* trace_dev, double dt, const double* dx,
                         int n_steps, void* stream);
/* host AoS convenience (allocates/frees HBM internally): one periodic step */
int exa_dg_step_host(exa_dg_plan* plan, double* u_host, double dt, const double* dx, int n_steps);

#ifdef __cplusplus
}
#endif
#endif
