/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C fp64 CPU restatement of the hot path of xdslproject/ExaHyPE for the
 * MI355X build.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (exahype_amd/) never does.
 *
 * What is pinned and what is not
 * ------------------------------
 *  - orc_flux_ref2d / orc_maxeig_ref2d / orc_fv_rusanov_faithful restate
 *    `Unit test/Functions.cpp:9-62` and `Unit test/test.cpp:3-111` of the
 *    reference statement for statement (2-D build: `Dimensions` undefined).
 *    Pinned against the compiled reference (oracle/_ref, built by
 *    oracle/Makefile from the sources where they lie) on the reference's
 *    well-defined outputs (the reference reads uninitialised heap
 *    elsewhere) and against tests/golden/fv_ref2d_*.json.
 *  - orc_fv_rusanov (corrected Rusanov with dt/h, all n_real variables) and
 *    every orc_aderdg_* function have NO counterpart in /root/reference
 *    (SURVEY.md F2, Appendix B): "parity unpinned".  They follow SURVEY.md
 *    Appendix A and are pinned by the known-answer tests of A.5 via the
 *    independent numpy restatement oracle/aderdg_numpy.py.
 *
 * Layouts (host, reference layout `CPPPrinter.py:247-261`): AoS, row-major,
 * variable fastest.  FV: Q[patch][i][j]([k])[var] incl. halo.  DG:
 * u[cell][node][var], cell = (cx*ny+cy)*nz+cz, node = (i*N+j)*N+k; axis 0 is
 * the reference's `i` (normal = 0).
 */
#include <omp.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GAMMA 1.4

enum { PDE_EULER_REF2D = 0, PDE_EULER = 1, PDE_ADVECTION = 2 };

/* ---- point-wise PDE terms ------------------------------------------------ */

/* Functions.cpp:9-37 compiled without `Dimensions` (2-D branch): reads Q[0..3],
 * writes F[0..3]; F[4..] untouched. */
void orc_flux_ref2d(const double* Q, int normal, double* F) {
    const double rho = Q[0], u = Q[1], v = Q[2], e = Q[3];
    const double irho = 1.0 / rho;
    const double p = (GAMMA - 1) * (e - 0.5 * irho * (u * u + v * v));
    const double coeff = irho * Q[normal + 1];
    F[0] = coeff * rho;
    F[1] = coeff * u;
    F[2] = coeff * v;
    F[3] = coeff * e + coeff * p;
    F[normal + 1] += p;
}

/* Functions.cpp:39-62, 2-D branch. */
double orc_maxeig_ref2d(const double* Q, int normal) {
    const double rho = Q[0], u = Q[1], v = Q[2], e = Q[3];
    const double irho = 1.0 / fabs(rho);
    const double p = (GAMMA - 1) * (e - 0.5 * irho * (u * u + v * v));
    const double c = sqrt(GAMMA * fabs(p) * irho);
    const double u_n = Q[normal + 1] * irho;
    return fmax(fabs(u_n - c), fabs(u_n + c));
}

/* Same arithmetic with a 3-component momentum (rho, m0, m1, m2, E); this is
 * Functions.cpp's 3-D branch without the stray F[3] overwrite (SURVEY B-4). */
void orc_flux_euler(const double* Q, int normal, double* F) {
    const double rho = Q[0], u = Q[1], v = Q[2], w = Q[3], e = Q[4];
    const double irho = 1.0 / rho;
    const double p = (GAMMA - 1) * (e - 0.5 * irho * (u * u + v * v + w * w));
    const double coeff = irho * Q[normal + 1];
    F[0] = coeff * rho;
    F[1] = coeff * u;
    F[2] = coeff * v;
    F[3] = coeff * w;
    F[4] = coeff * e + coeff * p;
    F[normal + 1] += p;
}

double orc_maxeig_euler(const double* Q, int normal) {
    const double rho = Q[0], u = Q[1], v = Q[2], w = Q[3], e = Q[4];
    const double irho = 1.0 / fabs(rho);
    const double p = (GAMMA - 1) * (e - 0.5 * irho * (u * u + v * v + w * w));
    const double c = sqrt(GAMMA * fabs(p) * irho);
    const double u_n = Q[normal + 1] * irho;
    return fmax(fabs(u_n - c), fabs(u_n + c));
}

/* Linear advection of every variable with the fixed velocity below. */
static const double ADV_A[3] = {1.0, 0.5, -0.75};

static void pde_flux(int pde, int m, const double* Q, int normal, double* F) {
    switch (pde) {
    case PDE_EULER_REF2D: orc_flux_ref2d(Q, normal, F); break;
    case PDE_EULER: orc_flux_euler(Q, normal, F); break;
    default:
        for (int v = 0; v < m; v++) F[v] = ADV_A[normal] * Q[v];
    }
}

static double pde_maxeig(int pde, const double* Q, int normal) {
    switch (pde) {
    case PDE_EULER_REF2D: return orc_maxeig_ref2d(Q, normal);
    case PDE_EULER: return orc_maxeig_euler(Q, normal);
    default: return fabs(ADV_A[normal]);
    }
}

void orc_pde_flux(int pde, int m, const double* Q, int normal, double* F) { pde_flux(pde, m, Q, normal, F); }
double orc_pde_maxeig(int pde, const double* Q, int normal) { return pde_maxeig(pde, Q, normal); }

/* ---- Finite-Volume Rusanov patch update ----------------------------------- */

static long ipow(long b, int e) { long r = 1; while (e-- > 0) r *= b; return r; }

/* test.cpp:64-66 `max(double*,double*)` (Functions.cpp:64-66) */
static double max2(const double* a, const double* b) { return *a > *b ? *a : *b; }

/*
 * Faithful mode: the ten loop nests of `Unit test/test.cpp:11-104`, generalised
 * from (dim=2,P=4,H=1,5+5 vars,1 patch) to any dim in {2,3}, P, H, n_real,
 * n_aux, n_patches with the same range rule (directional statements: interior
 * along the normal axis, full range along the others; first copy full, last
 * copy interior), the same evaluation order, no dt/h factor, dissipation on
 * variable 0 only with the reference's sign.
 * Temporaries are ZERO-initialised where the reference leaves heap garbage.
 */
void orc_fv_rusanov_faithful(double* Q, double dt, int dim, int P, int H, int n_real, int n_aux,
                             long n_patches, int pde) {
    const int S = P + 2 * H, V = n_real + n_aux, m = n_real;
    const long vol = ipow(S, dim);
    double* Qc = (double*)calloc((size_t)vol * V, sizeof(double));
    double* fl[3];
    double* ei[3];
    for (int d = 0; d < dim; d++) {
        fl[d] = (double*)malloc((size_t)vol * m * sizeof(double));
        ei[d] = (double*)malloc((size_t)vol * sizeof(double));
    }
    long st[3]; /* cell strides per axis (in cells) */
    if (dim == 2) { st[0] = S; st[1] = 1; st[2] = 0; } else { st[0] = (long)S * S; st[1] = S; st[2] = 1; }
    for (long patch = 0; patch < n_patches; patch++) {
        double* Qp = Q + patch * vol * V;
        for (int d = 0; d < dim; d++) {
            memset(fl[d], 0, (size_t)vol * m * sizeof(double));
            memset(ei[d], 0, (size_t)vol * sizeof(double));
        }
        /* L1 test.cpp:11-19 */
        memcpy(Qc, Qp, (size_t)vol * V * sizeof(double));
        int lo[3], hi[3];
#define FOR_RANGE(body)                                                                  \
    for (int i = lo[0]; i < hi[0]; i++)                                                  \
        for (int j = lo[1]; j < hi[1]; j++)                                              \
            for (int k = (dim == 3 ? lo[2] : 0); k < (dim == 3 ? hi[2] : 1); k++) {      \
                const long c = i * st[0] + j * st[1] + k * st[2];                        \
                body                                                                     \
            }
#define DIR_RANGE(d)                                                                     \
    for (int a = 0; a < 3; a++) { lo[a] = 0; hi[a] = S; }                                \
    lo[d] = H; hi[d] = P + H;
        /* L2/L3 test.cpp:20-39 */
        for (int d = 0; d < dim; d++) {
            DIR_RANGE(d)
            FOR_RANGE(pde_flux(pde, m, &Qc[c * V], d, &fl[d][c * m]);)
        }
        /* L4/L5 test.cpp:40-59 */
        for (int d = 0; d < dim; d++) {
            DIR_RANGE(d)
            FOR_RANGE(ei[d][c] = pde_maxeig(pde, &Qc[c * V], d);)
        }
        /* L6/L7 test.cpp:60-77 */
        for (int d = 0; d < dim; d++) {
            DIR_RANGE(d)
            FOR_RANGE(for (int var = 0; var < m; var++) Qc[c * V + var] =
                          Qc[c * V + var] - 0.5 * fl[d][(c + st[d]) * m + var] + 0.5 * fl[d][(c - st[d]) * m + var];)
        }
        /* L8/L9 test.cpp:78-95 */
        for (int d = 0; d < dim; d++) {
            DIR_RANGE(d)
            FOR_RANGE(Qc[c * V] = 0.5 * dt * ((-Qp[(c + st[d]) * V] + Qp[c * V]) * max2(&ei[d][c + st[d]], &ei[d][c]) +
                                             (Qp[(c - st[d]) * V] - Qp[c * V]) * max2(&ei[d][c - st[d]], &ei[d][c])) +
                                Qc[c * V];)
        }
        /* L10 test.cpp:96-104 */
        for (int a = 0; a < 3; a++) { lo[a] = H; hi[a] = P + H; }
        FOR_RANGE(for (int var = 0; var < V; var++) Qp[c * V + var] = Qc[c * V + var];)
    }
    free(Qc);
    for (int d = 0; d < dim; d++) { free(fl[d]); free(ei[d]); }
}

/*
 * Corrected mode (no counterpart in the reference -> unpinned):
 *   Q_c <- Q_c - (dt/h) sum_d (F*_{c+1/2,d} - F*_{c-1/2,d}),
 *   F*_{c+1/2} = 1/2 (f_d(Q_c)+f_d(Q_{c+1})) - 1/2 max(l_d(Q_c),l_d(Q_{c+1})) (Q_{c+1}-Q_c)
 * for all n_real variables; auxiliary variables and the halo are untouched.
 * Needs H >= 1.
 */
void orc_fv_rusanov(double* Q, double dt, double h, int dim, int P, int H, int n_real, int n_aux,
                    long n_patches, int pde) {
    const int S = P + 2 * H, V = n_real + n_aux, m = n_real;
    const long vol = ipow(S, dim);
    long st[3];
    if (dim == 2) { st[0] = S; st[1] = 1; st[2] = 0; } else { st[0] = (long)S * S; st[1] = S; st[2] = 1; }
#pragma omp parallel
    {
        double* Qold = (double*)malloc((size_t)vol * V * sizeof(double));
#pragma omp for schedule(static)
        for (long patch = 0; patch < n_patches; patch++) {
            double* Qp = Q + patch * vol * V;
            memcpy(Qold, Qp, (size_t)vol * V * sizeof(double));
            for (int i = H; i < P + H; i++)
                for (int j = H; j < P + H; j++)
                    for (int k = (dim == 3 ? H : 0); k < (dim == 3 ? P + H : 1); k++) {
                        const long c = i * st[0] + j * st[1] + k * st[2];
                        double acc[16], Fc[16], Fn[16];
                        for (int v = 0; v < m; v++) acc[v] = 0.0;
                        for (int d = 0; d < dim; d++) {
                            const double* qc = &Qold[c * V];
                            const double* qp = &Qold[(c + st[d]) * V];
                            const double* qm = &Qold[(c - st[d]) * V];
                            const double lc = pde_maxeig(pde, qc, d);
                            const double sp = fmax(lc, pde_maxeig(pde, qp, d));
                            const double sm = fmax(pde_maxeig(pde, qm, d), lc);
                            for (int v = 0; v < m; v++) Fc[v] = 0.0;
                            pde_flux(pde, m, qc, d, Fc);
                            for (int v = 0; v < m; v++) Fn[v] = 0.0;
                            pde_flux(pde, m, qp, d, Fn);
                            for (int v = 0; v < m; v++) acc[v] += 0.5 * (Fc[v] + Fn[v]) - 0.5 * sp * (qp[v] - qc[v]);
                            for (int v = 0; v < m; v++) Fn[v] = 0.0;
                            pde_flux(pde, m, qm, d, Fn);
                            for (int v = 0; v < m; v++) acc[v] -= 0.5 * (Fn[v] + Fc[v]) - 0.5 * sm * (qc[v] - qm[v]);
                        }
                        for (int v = 0; v < m; v++) Qp[c * V + v] = Qold[c * V + v] - dt / h * acc[v];
                    }
        }
        free(Qold);
    }
}

/* ---- ADER-DG (no counterpart in the reference) ---------- */

typedef struct {
    int dim, N, m, pde, n_it;
    long ncell[3];
    const double *w, *D, *Kxi, *phiL, *phiR, *iK1, *F0;
} dg_ctx;

/* out[.., i, ..] = sum_j M[i][j] in[.., j, ..] along an axis with stride `str`
 * and `outer`/`inner` slabs; block = values per entry (variables). */
static void apply_axis(const double* M, int N, const double* in, double* out, long outer, long str, int accumulate,
                       double scale) {
    /* array viewed as [outer][N][str] */
    for (long o = 0; o < outer; o++)
        for (int i = 0; i < N; i++)
            for (long s = 0; s < str; s++) {
                double acc = 0.0;
                for (int j = 0; j < N; j++) acc += M[i * N + j] * in[(o * N + j) * str + s];
                if (accumulate) out[(o * N + i) * str + s] += scale * acc;
                else out[(o * N + i) * str + s] = scale * acc;
            }
}

/*
 * Space-time predictor for ONE cell (A.2):
 *   q^(0)_l = u;  R_l = F0[l] u - dt w_l sum_d (1/dx_d) D_d f_d(q_l);  q <- iK1 R.
 * q layout [l][node][var]; nn = N^dim nodes.
 */
static void predictor_cell(const dg_ctx* c, const double* u, double dt, const double* dx, double* q, double* R,
                           double* F) {
    const int N = c->N, m = c->m, dim = c->dim;
    const long nn = ipow(N, dim);
    for (int l = 0; l < N; l++) memcpy(q + (long)l * nn * m, u, (size_t)nn * m * sizeof(double));
    for (int it = 0; it < c->n_it; it++) {
        for (int l = 0; l < N; l++) {
            double* Rl = R + (long)l * nn * m;
            for (long i = 0; i < nn * m; i++) Rl[i] = 0.0;
            for (int d = 0; d < dim; d++) {
                for (long x = 0; x < nn; x++) pde_flux(c->pde, m, q + ((long)l * nn + x) * m, d, F + x * m);
                const long str = ipow(N, dim - 1 - d) * m, outer = ipow(N, d);
                apply_axis(c->D, N, F, Rl, outer, str, 1, 1.0 / dx[d]);
            }
            for (long i = 0; i < nn * m; i++) Rl[i] = c->F0[l] * u[i] - dt * c->w[l] * Rl[i];
        }
        /* q[l'] = sum_l iK1[l'][l] R[l] : time axis is the outermost, stride nn*m */
        apply_axis(c->iK1, N, R, q, 1, nn * m, 0, 1.0);
    }
}

void orc_aderdg_predictor(int dim, int N, int m, int pde, int n_it, const double* ops, const double* u, double dt,
                          const double* dx, double* q);

/* ops packing: [w(N) D(N*N) Kxi(N*N) phiL(N) phiR(N) iK1(N*N) F0(N)] */
static void ctx_from_ops(dg_ctx* c, int dim, int N, int m, int pde, int n_it, const double* ops) {
    c->dim = dim; c->N = N; c->m = m; c->pde = pde; c->n_it = n_it;
    c->w = ops; c->D = c->w + N; c->Kxi = c->D + N * N; c->phiL = c->Kxi + N * N; c->phiR = c->phiL + N;
    c->iK1 = c->phiR + N; c->F0 = c->iK1 + N * N;
}

void orc_aderdg_predictor(int dim, int N, int m, int pde, int n_it, const double* ops, const double* u, double dt,
                          const double* dx, double* q) {
    dg_ctx c; ctx_from_ops(&c, dim, N, m, pde, n_it, ops);
    const long nn = ipow(N, dim);
    double* R = (double*)malloc((size_t)N * nn * m * sizeof(double));
    double* F = (double*)malloc((size_t)nn * m * sizeof(double));
    predictor_cell(&c, u, dt, dx, q, R, F);
    free(R); free(F);
}

/*
 * Stage A for every cell: predictor (n_it Picard iterations; n_it = 0 gives the
 * single-stage variant qbar := u, Fbar := f(u)), time averages, volume integral
 * (A.3) and face extrapolation.  u -> ustar (separate array) and
 * trace[((d*2+side)*ncells + cell)*(2*m*Nf) + (field*m+v)*Nf + y],
 * field 0 = qbar, field 1 = Fbar_d, side 0 = L (xi=0), 1 = R (xi=1),
 * y = lexicographic index of the remaining node axes.
 */
void orc_aderdg_stage_a(int dim, int N, int m, int pde, int n_it, const double* ops, long ncells, const double* u,
                        double dt, const double* dx, double* ustar, double* trace) {
    dg_ctx c; ctx_from_ops(&c, dim, N, m, pde, n_it, ops);
    const long nn = ipow(N, dim), Nf = ipow(N, dim - 1);
#pragma omp parallel
    {
        double* q = (double*)malloc((size_t)N * nn * m * sizeof(double));
        double* R = (double*)malloc((size_t)N * nn * m * sizeof(double));
        double* F = (double*)malloc((size_t)nn * m * sizeof(double));
        double* qbar = (double*)malloc((size_t)nn * m * sizeof(double));
        double* Fbar = (double*)malloc((size_t)nn * m * sizeof(double));
#pragma omp for schedule(static)
        for (long cell = 0; cell < ncells; cell++) {
            const double* uc = u + cell * nn * m;
            double* us = ustar + cell * nn * m;
            if (n_it > 0) {
                predictor_cell(&c, uc, dt, dx, q, R, F);
                for (long i = 0; i < nn * m; i++) {
                    double a = 0.0;
                    for (int l = 0; l < N; l++) a += c.w[l] * q[(long)l * nn * m + i];
                    qbar[i] = a;
                }
            } else {
                memcpy(qbar, uc, (size_t)nn * m * sizeof(double));
            }
            memcpy(us, uc, (size_t)nn * m * sizeof(double));
            for (int d = 0; d < dim; d++) {
                for (long i = 0; i < nn * m; i++) Fbar[i] = 0.0;
                if (n_it > 0) {
                    for (int l = 0; l < N; l++) {
                        for (long x = 0; x < nn; x++) pde_flux(c.pde, m, q + ((long)l * nn + x) * m, d, F + x * m);
                        for (long i = 0; i < nn * m; i++) Fbar[i] += c.w[l] * F[i];
                    }
                } else {
                    for (long x = 0; x < nn; x++) pde_flux(c.pde, m, uc + x * m, d, Fbar + x * m);
                }
                const long str = ipow(N, dim - 1 - d), outer = ipow(N, d);
                /* volume: us += dt/dx_d * (1/w_i) * sum_j Kxi[i][j] Fbar[j] */
                for (long o = 0; o < outer; o++)
                    for (int i = 0; i < N; i++)
                        for (long s = 0; s < str; s++)
                            for (int v = 0; v < m; v++) {
                                double a = 0.0;
                                for (int j = 0; j < N; j++) a += c.Kxi[i * N + j] * Fbar[((o * N + j) * str + s) * m + v];
                                us[((o * N + i) * str + s) * m + v] += dt / dx[d] * a / c.w[i];
                            }
                /* traces */
                for (int side = 0; side < 2; side++) {
                    const double* phi = side ? c.phiR : c.phiL;
                    double* tr = trace + (((long)d * 2 + side) * ncells + cell) * (2 * m * Nf);
                    for (long o = 0; o < outer; o++)
                        for (long s = 0; s < str; s++) {
                            const long y = o * str + s;
                            for (int v = 0; v < m; v++) {
                                double aq = 0.0, aF = 0.0;
                                for (int j = 0; j < N; j++) {
                                    aq += phi[j] * qbar[((o * N + j) * str + s) * m + v];
                                    aF += phi[j] * Fbar[((o * N + j) * str + s) * m + v];
                                }
                                tr[(0 * m + v) * Nf + y] = aq;
                                tr[(1 * m + v) * Nf + y] = aF;
                            }
                        }
                }
            }
        }
        free(q); free(R); free(F); free(qbar); free(Fbar);
    }
}

/*
 * Stage B for every cell of a PERIODIC nc[0] x nc[1] (x nc[2]) grid: Rusanov
 * flux on both faces per direction (A.4; s = max over the face's nodes of both
 * sides' maxEigenvalue) and the surface corrector.  ustar -> unew.
 */
void orc_aderdg_stage_b(int dim, int N, int m, int pde, const double* ops, const long* nc, const double* ustar,
                        const double* trace, double dt, const double* dx, double* unew) {
    dg_ctx c; ctx_from_ops(&c, dim, N, m, pde, 0, ops);
    const long nn = ipow(N, dim), Nf = ipow(N, dim - 1);
    long ncells = 1;
    for (int d = 0; d < dim; d++) ncells *= nc[d];
    long cst[3] = {0, 0, 0};
    { long s = 1; for (int d = dim - 1; d >= 0; d--) { cst[d] = s; s *= nc[d]; } }
#pragma omp parallel
    {
        double* Fs = (double*)malloc((size_t)2 * Nf * m * sizeof(double));
        double qa[16], qb[16];
#pragma omp for schedule(static)
        for (long cell = 0; cell < ncells; cell++) {
            const double* us = ustar + cell * nn * m;
            double* un = unew + cell * nn * m;
            memcpy(un, us, (size_t)nn * m * sizeof(double));
            for (int d = 0; d < dim; d++) {
                const long cd = (cell / cst[d]) % nc[d];
                const long left = cell + ((cd == 0 ? nc[d] - 1 : cd - 1) - cd) * cst[d];
                const long right = cell + ((cd == nc[d] - 1 ? 0 : cd + 1) - cd) * cst[d];
                for (int face = 0; face < 2; face++) {
                    /* face 0: between `left` (its R trace, "-") and this cell (its L trace, "+");
                     * face 1: between this cell (R, "-") and `right` (L, "+"). */
                    const long cm = face ? cell : left, cp = face ? right : cell;
                    const double* tm = trace + (((long)d * 2 + 1) * ncells + cm) * (2 * m * Nf);
                    const double* tp = trace + (((long)d * 2 + 0) * ncells + cp) * (2 * m * Nf);
                    double s = 0.0;
                    for (long y = 0; y < Nf; y++) {
                        for (int v = 0; v < m; v++) { qa[v] = tm[v * Nf + y]; qb[v] = tp[v * Nf + y]; }
                        s = fmax(s, fmax(pde_maxeig(c.pde, qa, d), pde_maxeig(c.pde, qb, d)));
                    }
                    for (long y = 0; y < Nf; y++)
                        for (int v = 0; v < m; v++)
                            Fs[((long)face * Nf + y) * m + v] = 0.5 * (tm[(m + v) * Nf + y] + tp[(m + v) * Nf + y]) -
                                                                 0.5 * s * (tp[v * Nf + y] - tm[v * Nf + y]);
                }
                const long str = ipow(N, dim - 1 - d), outer = ipow(N, d);
                for (long o = 0; o < outer; o++)
                    for (int i = 0; i < N; i++)
                        for (long sidx = 0; sidx < str; sidx++) {
                            const long y = o * str + sidx;
                            for (int v = 0; v < m; v++)
                                un[((o * N + i) * str + sidx) * m + v] -=
                                    dt / dx[d] / c.w[i] *
                                    (c.phiR[i] * Fs[((long)1 * Nf + y) * m + v] - c.phiL[i] * Fs[((long)0 * Nf + y) * m + v]);
                        }
            }
        }
        free(Fs);
    }
}

/* One full step on a periodic grid, in place on u (allocates ustar + traces). */
void orc_aderdg_step(int dim, int N, int m, int pde, int n_it, const double* ops, const long* nc, double* u, double dt,
                     const double* dx) {
    const long nn = ipow(N, dim), Nf = ipow(N, dim - 1);
    long ncells = 1;
    for (int d = 0; d < dim; d++) ncells *= nc[d];
    double* us = (double*)malloc((size_t)ncells * nn * m * sizeof(double));
    double* tr = (double*)malloc((size_t)dim * 2 * ncells * 2 * m * Nf * sizeof(double));
    orc_aderdg_stage_a(dim, N, m, pde, n_it, ops, ncells, u, dt, dx, us, tr);
    orc_aderdg_stage_b(dim, N, m, pde, ops, nc, us, tr, dt, dx, u);
    free(us); free(tr);
}

/* OpenMP team size of the restatement's parallel loops (bench.py's CPU-baseline leg: one thread per usable core). */
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_get_max_threads(void) { return omp_get_max_threads(); }

/* n_steps full steps on a periodic grid, in place on u; ustar + traces are allocated ONCE (bench.py's CPU-baseline
 * leg: no allocation inside the timed steps).  Same arithmetic as n_steps calls of orc_aderdg_step. */
void orc_aderdg_run(int dim, int N, int m, int pde, int n_it, const double* ops, const long* nc, double* u, double dt,
                    const double* dx, int n_steps) {
    const long nn = ipow(N, dim), Nf = ipow(N, dim - 1);
    long ncells = 1;
    for (int d = 0; d < dim; d++) ncells *= nc[d];
    double* us = (double*)malloc((size_t)ncells * nn * m * sizeof(double));
    double* tr = (double*)malloc((size_t)dim * 2 * ncells * 2 * m * Nf * sizeof(double));
    for (int s = 0; s < n_steps; s++) {
        orc_aderdg_stage_a(dim, N, m, pde, n_it, ops, ncells, u, dt, dx, us, tr);
        orc_aderdg_stage_b(dim, N, m, pde, ops, nc, us, tr, dt, dx, u);
    }
    free(us); free(tr);
}
