// C-ABI of libexahype_hip.so (include/exahype_hip.h).  Host-side only: argument
// checking, plan bookkeeping, staging for the *_host convenience calls, and
// dispatch into the kernel units.  Never throws; every failure sets the
// thread-local message behind exa_last_error().
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <dlfcn.h>
#include "../../include/exahype_hip.h"
#include "exa_launch.hpp"

namespace exa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define EXA_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return EXA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

const DgLaunchTable *dg_table_2_0(), *dg_table_2_1(), *dg_table_2_2(), *dg_table_3_1(), *dg_table_3_2();

// ---- run-time registered PDE term sets (side libraries generated from SymPy expressions) -------------
struct UserPde {
    void* handle;
    const DgLaunchTable* dg[4];
    int nv;
    int flags;             // EXA_PDE_FLAG_*
    int (*fv)(int, int, int, int, int, int, long, double*, double, double, const long*, void*, double*, const double*, double, const void*);
    int (*fvmax)(int, int, int, int, int, long, const double*, double*, void*, const double*, double, double);
    int (*ev)(int, long, int, const double*, double*, double*, void*, const double*, double);
};
static std::vector<UserPde> g_user;

int user_fv_launch(int pde, int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, double* Q, double dt,
                   double h, const long* slot, hipStream_t s, double* out, const double* centre, double t, const FvGridArgs* grid) {
    if (pde - 100 >= (int)g_user.size() || !g_user[pde - 100].fv) { set_error("pde %d is not registered", pde); return -1; }
    return g_user[pde - 100].fv(mode, dim, P, H, n_real, n_aux, n_patches, Q, dt, h, slot, (void*)s, out, centre, t, grid);
}
int user_fv_maxeig(int pde, int dim, int P, int H, int n_real, int n_aux, long n_patches, const double* Q, double* lam, hipStream_t s,
                   const double* centre, double t, double h) {
    if (pde - 100 >= (int)g_user.size() || !g_user[pde - 100].fvmax) { set_error("pde %d is not registered", pde); return -1; }
    return g_user[pde - 100].fvmax(dim, P, H, n_real, n_aux, n_patches, Q, lam, (void*)s, centre, t, h);
}
int user_pde_eval(int pde, int normal, long n, int stride, const double* Q, double* F, double* lam, hipStream_t s, const double* X, double t) {
    if (pde - 100 >= (int)g_user.size() || !g_user[pde - 100].ev) { set_error("pde %d is not registered", pde); return -1; }
    return g_user[pde - 100].ev(normal, n, stride, Q, F, lam, (void*)s, X, t);
}

const DgLaunchTable* dg_launch_table(int dim, int pde) {
    if (pde >= 100) return (pde - 100 < (int)g_user.size() && dim >= 2 && dim <= 3) ? g_user[pde - 100].dg[dim] : nullptr;
    if (dim == 2 && pde == 0) return dg_table_2_0();
    if (dim == 2 && pde == 1) return dg_table_2_1();
    if (dim == 2 && pde == 2) return dg_table_2_2();
    if (dim == 3 && pde == 1) return dg_table_3_1();
    if (dim == 3 && pde == 2) return dg_table_3_2();
    return nullptr;
}

static long lpow(long b, int e) { long r = 1; while (e-- > 0) r *= b; return r; }

}  // namespace exa

using namespace exa;

struct exa_fv_plan {
    int device, mode, dim, P, H, n_real, n_aux, pde;
    long n_patches, count;
};

struct exa_dg_plan {
    int device, dim, N, nv, pde, n_it;
    long nc[3], ncells;
    const DgLaunchTable* tab;
    DgOpsHost ops;
};

extern "C" {

int exa_version(void) { return 100; }

const char* exa_last_error(void) { return g_err; }

int exa_device_count(int* count) {
    if (!count) { set_error("exa_device_count: NULL argument"); return EXA_ERR_INVALID; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return EXA_ERR_NO_DEVICE; }
    *count = n;
    return EXA_OK;
}

static int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libexahype_hip has no CPU fallback", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return EXA_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return EXA_ERR_INVALID; }
    EXA_HIP(hipSetDevice(device));
    return EXA_OK;
}

int exa_register_pde(const char* library_path, int* pde_id) {
    if (!library_path || !pde_id) { set_error("exa_register_pde: NULL argument"); return EXA_ERR_INVALID; }
    void* h = dlopen(library_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error("exa_register_pde: %s", dlerror()); return EXA_ERR_INVALID; }
    UserPde u{};
    u.handle = h;
    typedef const void* (*tab_fn)();
    for (int d = 2; d <= 3; d++) {
        char name[64];
        snprintf(name, sizeof(name), "exa_user_dg_table_%d", d);
        tab_fn f = (tab_fn)dlsym(h, name);
        u.dg[d] = f ? static_cast<const DgLaunchTable*>(f()) : nullptr;
    }
    int (*nvf)() = (int (*)())dlsym(h, "exa_user_nv");
    u.fv = (decltype(u.fv))dlsym(h, "exa_user_fv_launch");
    u.fvmax = (decltype(u.fvmax))dlsym(h, "exa_user_fv_maxeig");
    u.ev = (decltype(u.ev))dlsym(h, "exa_user_pde_eval");
    if (!nvf || (!u.dg[2] && !u.dg[3] && !u.fv)) { dlclose(h); set_error("%s exports no exahype_amd PDE entry points", library_path); return EXA_ERR_INVALID; }
    u.nv = nvf();
    int (*flf)() = (int (*)())dlsym(h, "exa_user_pde_flags");
    u.flags = flf ? flf() : 0;
    g_user.push_back(u);
    *pde_id = 100 + (int)g_user.size() - 1;
    return EXA_OK;
}

int exa_pde_flags(int pde) {
    if (pde >= 100) return (pde - 100 < (int)g_user.size()) ? g_user[pde - 100].flags : 0;
    return 0;
}

int exa_pde_eval_device(int pde, int normal, long n, int stride, const double* Q_dev, double* F_dev, double* lambda_dev,
                        void* stream) {
    return exa_pde_eval_device_at(pde, normal, n, stride, Q_dev, nullptr, 0.0, F_dev, lambda_dev, stream);
}

int exa_pde_eval_device_at(int pde, int normal, long n, int stride, const double* Q_dev, const double* x_dev, double t, double* F_dev,
                           double* lambda_dev, void* stream) {
    if (pde >= 100) {
        if (normal < 0 || normal > 2 || n < 0 || stride < 1 || !Q_dev) { set_error("exa_pde_eval_device: bad argument"); return EXA_ERR_INVALID; }
        return pde_eval_launch(pde, normal, n, stride, Q_dev, F_dev, lambda_dev, (hipStream_t)stream, x_dev, t);
    }
    if (pde < 0 || pde > 2 || normal < 0 || normal > 2 || n < 0 || stride < 1 || !Q_dev) {
        set_error("exa_pde_eval_device: bad argument");
        return EXA_ERR_INVALID;
    }
    if (pde == EXA_PDE_EULER_REF2D && (normal > 1 || stride < 4)) { set_error("EULER_REF2D needs normal < 2 and stride >= 4"); return EXA_ERR_INVALID; }
    if (pde == EXA_PDE_EULER && stride < 5) { set_error("EULER needs stride >= 5"); return EXA_ERR_INVALID; }
    return pde_eval_launch(pde, normal, n, stride, Q_dev, F_dev, lambda_dev, (hipStream_t)stream, x_dev, t);
}

/* ---- FV ---------------------------------------------------------------------- */

int exa_fv_plan_create(int device, int mode, int dim, int patch_size, int halo_size, int n_real, int n_aux,
                       long n_patches, int pde, exa_fv_plan** plan) {
    if (!plan) { set_error("exa_fv_plan_create: NULL plan"); return EXA_ERR_INVALID; }
    *plan = nullptr;
    // same viability rule as the reference's KernelBuilder (exahype/KernelBuilder.py:41-48) ...
    if ((dim != 2 && dim != 3) || patch_size < 1 || halo_size < 0) { set_error("check viability of inputs"); return EXA_ERR_INVALID; }
    // ... plus what the 3-point stencil itself needs
    if (halo_size < 1) { set_error("the Rusanov stencil reads one halo layer: halo_size must be >= 1"); return EXA_ERR_INVALID; }
    if (mode != EXA_FV_FAITHFUL && mode != EXA_FV_RUSANOV) { set_error("unknown FV mode %d", mode); return EXA_ERR_INVALID; }
    if (n_real < 1 || n_real > 8 || n_aux < 0 || n_patches < 0) { set_error("n_real must be 1..8, n_aux >= 0, n_patches >= 0"); return EXA_ERR_INVALID; }
    if (pde == EXA_PDE_EULER_REF2D && (dim != 2 || n_real < 4)) { set_error("EULER_REF2D is the reference's 2-D term set (n_real >= 4)"); return EXA_ERR_INVALID; }
    if (pde == EXA_PDE_EULER && n_real < 5) { set_error("EULER needs n_real >= 5"); return EXA_ERR_INVALID; }
    if (pde >= 100) {
        if (pde - 100 >= (int)g_user.size()) { set_error("pde %d is not registered", pde); return EXA_ERR_INVALID; }
        if (n_real < g_user[pde - 100].nv) { set_error("pde %d evolves %d variables; n_real = %d", pde, g_user[pde - 100].nv, n_real); return EXA_ERR_INVALID; }
    } else if (pde < 0 || pde > 2) { set_error("unknown pde %d", pde); return EXA_ERR_INVALID; }
    const long ncell = lpow(patch_size, dim);
    if (ncell > 4096) { set_error("FV patch with %ld volumes exceeds the 4096 a workgroup keeps in registers", ncell); return EXA_ERR_INVALID; }
    int rc = use_device(device);
    if (rc) return rc;
    exa_fv_plan* p = new (std::nothrow) exa_fv_plan;
    if (!p) { set_error("out of host memory"); return EXA_ERR_ALLOC; }
    p->device = device; p->mode = mode; p->dim = dim; p->P = patch_size; p->H = halo_size;
    p->n_real = n_real; p->n_aux = n_aux; p->pde = pde; p->n_patches = n_patches;
    p->count = n_patches * lpow(patch_size + 2 * halo_size, dim) * (n_real + n_aux);
    *plan = p;
    return EXA_OK;
}

int exa_fv_plan_destroy(exa_fv_plan* plan) { delete plan; return EXA_OK; }

long exa_fv_q_count(const exa_fv_plan* plan) { return plan ? plan->count : 0; }

int exa_fv_time_step_device_masked(exa_fv_plan* p, double* Q_dev, const long* slot_dev, double dt, double h, void* stream) {
    if (!p || (!Q_dev && p->count > 0)) { set_error("exa_fv_time_step_device: NULL argument"); return EXA_ERR_INVALID; }
    if (p->mode == EXA_FV_RUSANOV && !(h > 0.0)) { set_error("EXA_FV_RUSANOV needs the volume size h > 0"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    return fv_launch(p->mode, p->dim, p->P, p->H, p->n_real, p->n_aux, p->n_patches, p->pde, Q_dev, dt, h, slot_dev, (hipStream_t)stream);
}

int exa_fv_time_step_device_masked_at(exa_fv_plan* p, double* Q_dev, const long* slot_dev, const double* centre_dev, double t, double dt, double h,
                                      void* stream) {
    if (!p || (!Q_dev && p->count > 0)) { set_error("exa_fv_time_step_device_masked_at: NULL argument"); return EXA_ERR_INVALID; }
    if (!(h > 0.0)) { set_error("exa_fv_time_step_device_masked_at needs the volume size h > 0 (volume centres, dt / h)"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    return fv_launch(p->mode, p->dim, p->P, p->H, p->n_real, p->n_aux, p->n_patches, p->pde, Q_dev, dt, h, slot_dev, (hipStream_t)stream, nullptr,
                     centre_dev, t);
}

int exa_fv_time_step_device_oop(exa_fv_plan* p, const double* QIn_dev, double* QOut_dev, const double* centre_dev, double t, double dt,
                                double h, void* stream) {
    if (!p || ((!QIn_dev || !QOut_dev) && p->count > 0)) { set_error("exa_fv_time_step_device_oop: NULL argument"); return EXA_ERR_INVALID; }
    if (!(h > 0.0)) { set_error("exa_fv_time_step_device_oop needs the volume size h > 0 (volume centres, dt / h)"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    // (the kernel only reads QIn: the const is cast away for the signature it shares with the in-place call)
    return fv_launch(p->mode, p->dim, p->P, p->H, p->n_real, p->n_aux, p->n_patches, p->pde, const_cast<double*>(QIn_dev), dt, h, nullptr,
                     (hipStream_t)stream, QOut_dev, centre_dev, t);
}

int exa_fv_time_step_device_at(exa_fv_plan* p, double* Q_dev, const double* centre_dev, double t, double dt, double h, void* stream) {
    if (!p || (!Q_dev && p->count > 0)) { set_error("exa_fv_time_step_device_at: NULL argument"); return EXA_ERR_INVALID; }
    if (!(h > 0.0)) { set_error("exa_fv_time_step_device_at needs the volume size h > 0 (volume centres, dt / h)"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    return fv_launch(p->mode, p->dim, p->P, p->H, p->n_real, p->n_aux, p->n_patches, p->pde, Q_dev, dt, h, nullptr, (hipStream_t)stream, nullptr,
                     centre_dev, t);
}

int exa_fv_grid_step_device(exa_fv_plan* p, const double* Q_dev, double* QNext_dev, const long* grid, const double* boundary_dev,
                            const double* centre_dev, double t, double dt, double h, double* lambda_next_dev, void* stream) {
    if (!p || !grid || ((!Q_dev || !QNext_dev) && p->count > 0)) { set_error("exa_fv_grid_step_device: NULL argument"); return EXA_ERR_INVALID; }
    if (Q_dev == QNext_dev) { set_error("exa_fv_grid_step_device: the new states need an array of their own (the neighbours read the old ones)"); return EXA_ERR_INVALID; }
    if (p->mode == EXA_FV_RUSANOV && !(h > 0.0)) { set_error("EXA_FV_RUSANOV needs the volume size h > 0"); return EXA_ERR_INVALID; }
    FvGridArgs ga{QNext_dev, boundary_dev, {1, 1, 1}, lambda_next_dev};
    long n = 1;
    for (int a = 0; a < p->dim; a++) {
        if (grid[a] < 1 || grid[a] > 0x7fffffffL) { set_error("exa_fv_grid_step_device: grid[%d] = %ld", a, grid[a]); return EXA_ERR_INVALID; }
        ga.g[a] = (int)grid[a];
        n *= grid[a];
    }
    if (n != p->n_patches) { set_error("exa_fv_grid_step_device: the grid has %ld patches, the plan %ld", n, p->n_patches); return EXA_ERR_INVALID; }
    // (the kernels decode a patch's grid coordinates in 32-bit arithmetic: fv_grid_coords)
    if (n > 0xffffffffL) { set_error("exa_fv_grid_step_device: %ld patches -- a grid holds at most 2^32 - 1", n); return EXA_ERR_INVALID; }
    if (p->H > p->P) { set_error("exa_fv_grid_step_device: halo_size %d exceeds patch_size %d (the halo would reach past the face neighbour)", p->H, p->P); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    return fv_launch(p->mode, p->dim, p->P, p->H, p->n_real, p->n_aux, p->n_patches, p->pde, const_cast<double*>(Q_dev), dt, h, nullptr,
                     (hipStream_t)stream, nullptr, centre_dev, t, &ga);
}

int exa_fv_max_eigenvalue(exa_fv_plan* p, const double* Q_dev, int halo_less, const double* centre_dev, double t, double h, double* lambda_dev,
                          void* stream) {
    if (!p || !lambda_dev || (!Q_dev && p->count > 0)) { set_error("exa_fv_max_eigenvalue: NULL argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    // (a halo-less array is a patch array with halo_size 0: every volume is an interior volume)
    return fv_maxeig_launch(p->dim, p->P, halo_less ? 0 : p->H, p->n_real, p->n_aux, p->n_patches, p->pde, Q_dev, lambda_dev, (hipStream_t)stream,
                            centre_dev, t, h);
}

long exa_fv_qout_count(const exa_fv_plan* plan) {
    return plan ? plan->n_patches * lpow(plan->P, plan->dim) * (plan->n_real + plan->n_aux) : 0;
}

int exa_fv_time_step_device(exa_fv_plan* p, double* Q_dev, double dt, double h, void* stream) {
    return exa_fv_time_step_device_masked(p, Q_dev, nullptr, dt, h, stream);
}

int exa_fv_time_step_host(exa_fv_plan* p, double* Q_host, double dt, double h) {
    if (!p || !Q_host) { set_error("exa_fv_time_step_host: NULL argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    if (p->count == 0) return EXA_OK;
    double* d = nullptr;
    const size_t bytes = (size_t)p->count * sizeof(double);
    EXA_HIP(hipMalloc(&d, bytes));
    hipError_t e = hipMemcpy(d, Q_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = exa_fv_time_step_device(p, d, dt, h, nullptr);
        if (rc == EXA_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(Q_host, d, bytes, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("exa_fv_time_step_host: %s", hipGetErrorString(e)); return EXA_ERR_HIP; }
    return EXA_OK;
}

/* ---- DG ---------------------------------------------------------------------- */

int exa_dg_plan_create(int device, int dim, int N, int n_vars, int pde, int n_picard, const long* ncells,
                       exa_dg_plan** plan) {
    if (!plan || !ncells) { set_error("exa_dg_plan_create: NULL argument"); return EXA_ERR_INVALID; }
    *plan = nullptr;
    if (dim != 2 && dim != 3) { set_error("check viability of inputs"); return EXA_ERR_INVALID; }
    const DgLaunchTable* tab = dg_launch_table(dim, pde);
    if (!tab) { set_error("ADER-DG: no kernels for dim %d, pde %d", dim, pde); return EXA_ERR_INVALID; }
    if (N < 2 || N > tab->max_n) {
        set_error("ADER-DG: N = %d unsupported for dim %d (2..%d)", N, dim, tab->max_n);
        return EXA_ERR_INVALID;
    }
    if (n_vars != tab->nv) { set_error("ADER-DG: pde %d evolves %d variables, got n_vars = %d", pde, tab->nv, n_vars); return EXA_ERR_INVALID; }
    long nc = 1;
    for (int d = 0; d < dim; d++) {
        if (ncells[d] < 1) { set_error("ncells[%d] = %ld", d, ncells[d]); return EXA_ERR_INVALID; }
        nc *= ncells[d];
    }
    int rc = use_device(device);
    if (rc) return rc;
    exa_dg_plan* p = new (std::nothrow) exa_dg_plan;
    if (!p) { set_error("out of host memory"); return EXA_ERR_ALLOC; }
    p->device = device; p->dim = dim; p->N = N; p->nv = n_vars; p->pde = pde;
    p->n_it = n_picard < 0 ? N : n_picard;
    p->nc[0] = ncells[0]; p->nc[1] = ncells[1]; p->nc[2] = dim == 3 ? ncells[2] : 1;
    p->ncells = nc;
    p->tab = tab;
    if (build_dg_operators(N, &p->ops) != 0) { delete p; set_error("operator construction failed for N = %d", N); return EXA_ERR_INVALID; }
    // operator block image in HBM
    p->ops.dev = nullptr;
    const size_t ob = tab->ops_image(N, &p->ops, nullptr);
    std::vector<char> img(ob);
    tab->ops_image(N, &p->ops, img.data());
    hipError_t e = hipMalloc(&p->ops.dev, ob);
    if (e == hipSuccess) e = hipMemcpy(p->ops.dev, img.data(), ob, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->ops.dev) (void)hipFree(p->ops.dev);
        delete p;
        set_error("operator block upload failed: %s", hipGetErrorString(e));
        return EXA_ERR_HIP;
    }
    p->ops.scratch = nullptr;
    p->ops.lim = nullptr;
    p->ops.stage_a_reserve = 0;
    p->ops.origin[0] = p->ops.origin[1] = p->ops.origin[2] = 0.0;
    p->ops.time = 0.0;
    p->ops.stage_a_variant = EXA_STAGE_A_AUTO;
    if (const char* ev = getenv("EXA_STAGE_A")) {
        if (!strcmp(ev, "lds")) p->ops.stage_a_variant = EXA_STAGE_A_LDS;
        else if (!strcmp(ev, "reg")) p->ops.stage_a_variant = EXA_STAGE_A_REG;
    }
    const size_t sb = tab->scratch_bytes(N);
    if (sb > 0) {
        e = hipMalloc(&p->ops.scratch, sb);
        if (e != hipSuccess) {
            (void)hipFree(p->ops.dev);
            delete p;
            set_error("scratch slab (%zu B) allocation failed: %s", sb, hipGetErrorString(e));
            return EXA_ERR_ALLOC;
        }
    }
    *plan = p;
    return EXA_OK;
}

int exa_dg_plan_destroy(exa_dg_plan* plan) {
    if (plan) {
        if (plan->ops.scratch) (void)hipFree(plan->ops.scratch);
        if (plan->ops.lim) (void)hipFree(plan->ops.lim);
        if (plan->ops.dev) (void)hipFree(plan->ops.dev);
        delete plan;
    }
    return EXA_OK;
}

int exa_dg_plan_set_stage_a(exa_dg_plan* plan, int variant) {
    if (!plan) { set_error("exa_dg_plan_set_stage_a: NULL plan"); return EXA_ERR_INVALID; }
    if (variant != EXA_STAGE_A_AUTO && variant != EXA_STAGE_A_LDS && variant != EXA_STAGE_A_REG) {
        set_error("exa_dg_plan_set_stage_a: variant %d (EXA_STAGE_A_AUTO|LDS|REG)", variant);
        return EXA_ERR_INVALID;
    }
    plan->ops.stage_a_variant = variant;
    return EXA_OK;
}

int exa_dg_plan_set_stage_a_reserve(exa_dg_plan* plan, int workgroups) {
    if (!plan || workgroups < 0) { set_error("exa_dg_plan_set_stage_a_reserve: bad argument"); return EXA_ERR_INVALID; }
    plan->ops.stage_a_reserve = workgroups;
    return EXA_OK;
}

const char* exa_dg_stage_a_kernel(const exa_dg_plan* plan) {
    if (!plan || !plan->tab->stage_a_name) return "";
    return plan->tab->stage_a_name(plan->N, plan->n_it, plan->ops.stage_a_variant);
}

long exa_dg_dof_count(const exa_dg_plan* p) { return p ? p->ncells * lpow(p->N, p->dim) * p->nv : 0; }
long exa_dg_trace_count(const exa_dg_plan* p) { return p ? (long)p->dim * 2 * p->ncells * 2 * p->nv * lpow(p->N, p->dim - 1) : 0; }
long exa_dg_face_count(const exa_dg_plan* p, int d) {
    if (!p || d < 0 || d >= p->dim) return 0;
    return (p->ncells / p->nc[d]) * 2 * p->nv * lpow(p->N, p->dim - 1);
}

int exa_dg_operators(const exa_dg_plan* p, double* xi, double* w, double* D, double* Kxi, double* phiL, double* phiR,
                     double* iK1) {
    if (!p) { set_error("exa_dg_operators: NULL plan"); return EXA_ERR_INVALID; }
    const int N = p->N;
    if (xi) memcpy(xi, p->ops.xi, N * sizeof(double));
    if (w) memcpy(w, p->ops.w, N * sizeof(double));
    if (D) memcpy(D, p->ops.D, N * N * sizeof(double));
    if (Kxi) memcpy(Kxi, p->ops.Kxi, N * N * sizeof(double));
    if (phiL) memcpy(phiL, p->ops.phiL, N * sizeof(double));
    if (phiR) memcpy(phiR, p->ops.phiR, N * sizeof(double));
    if (iK1) memcpy(iK1, p->ops.iK1, N * N * sizeof(double));
    return EXA_OK;
}

int exa_dg_work(const exa_dg_plan* p, double* fa, double* fb, double* ba, double* bb) {
    if (!p) { set_error("exa_dg_work: NULL plan"); return EXA_ERR_INVALID; }
    // c_F = c_lambda = 20 flop
    const double d = p->dim, m = p->nv, N = p->N, cF = 20, cL = 20;
    const double Nd = (double)lpow(p->N, p->dim), Nd1 = Nd * N, Nd2 = Nd1 * N, Nf = Nd / N;
    double a;
    if (p->n_it > 0) a = p->n_it * (2 * (d + 1) * m * Nd2 + d * cF * Nd1) + 2 * (d + 1) * m * Nd1 + 2 * d * m * Nd1 + 8 * d * m * Nd;
    else a = d * cF * Nd + 2 * d * m * Nd1 + 8 * d * m * Nd;
    const double b = d * Nf * (2 * cL + 6 * m) + 4 * d * m * Nd;
    if (fa) *fa = a * p->ncells;
    if (fb) *fb = b * p->ncells;
    if (ba) *ba = (16 * m * Nd + 32 * d * m * Nf) * p->ncells;   // u read + u* written + traces written
    if (bb) *bb = (16 * m * Nd + 64 * d * m * Nf) * p->ncells;   // u* read + u written + own and neighbour traces read
    return EXA_OK;
}

static void inv_dx(const exa_dg_plan* p, const double* dx, double* idx) {
    for (int d = 0; d < 3; d++) idx[d] = d < p->dim ? 1.0 / dx[d] : 0.0;
}

static int make_box(const exa_dg_plan* p, const long* lo, const long* hi, CellBox* box) {
    box->nbox = 1;
    for (int d = 0; d < 3; d++) {
        box->nc[d] = p->nc[d];
        box->lo[d] = (lo && d < p->dim) ? lo[d] : 0;
        const long h = (hi && d < p->dim) ? hi[d] : p->nc[d];
        if (box->lo[d] < 0 || h > p->nc[d] || h < box->lo[d]) {
            set_error("cell box [%ld,%ld) outside the block in direction %d", box->lo[d], h, d);
            return EXA_ERR_INVALID;
        }
        box->nb[d] = h - box->lo[d];
        box->nbox *= box->nb[d];
    }
    return EXA_OK;
}

int exa_dg_predictor_volume_box(exa_dg_plan* p, double* u_dev, double* trace_dev, const long* lo, const long* hi, double dt,
                                const double* dx, void* stream) {
    if (!p || !u_dev || !trace_dev || !dx) { set_error("exa_dg_predictor_volume: NULL argument"); return EXA_ERR_INVALID; }
    for (int d = 0; d < p->dim; d++)
        if (!(dx[d] > 0.0)) { set_error("dx[%d] must be > 0", d); return EXA_ERR_INVALID; }
    CellBox box;
    int rc = make_box(p, lo, hi, &box);
    if (rc) return rc;
    rc = use_device(p->device);
    if (rc) return rc;
    double idx[3];
    inv_dx(p, dx, idx);
    return p->tab->stage_a(p->N, u_dev, u_dev, trace_dev, p->ncells, &box, dt, idx, p->n_it, &p->ops, (hipStream_t)stream);
}

int exa_dg_predictor_volume(exa_dg_plan* p, double* u_dev, double* trace_dev, double dt, const double* dx, void* stream) {
    return exa_dg_predictor_volume_box(p, u_dev, trace_dev, nullptr, nullptr, dt, dx, stream);
}

static int riemann_corrector(exa_dg_plan* p, double* u_dev, const double* trace_dev, const double* const* ghost_dev,
                             const long* lo, const long* hi, double dt, const double* dx, double* lambda_dev, void* stream);

int exa_dg_riemann_corrector(exa_dg_plan* p, double* u_dev, const double* trace_dev, const double* const* ghost_dev,
                             const long* lo, const long* hi, double dt, const double* dx, void* stream) {
    return riemann_corrector(p, u_dev, trace_dev, ghost_dev, lo, hi, dt, dx, nullptr, stream);
}

int exa_dg_riemann_corrector_cfl(exa_dg_plan* p, double* u_dev, const double* trace_dev, const double* const* ghost_dev,
                                 const long* lo, const long* hi, double dt, const double* dx, double* lambda_dev, void* stream) {
    if (!lambda_dev) { set_error("exa_dg_riemann_corrector_cfl: NULL lambda_dev"); return EXA_ERR_INVALID; }
    if (p && (exa_pde_flags(p->pde) & EXA_PDE_FLAG_XT)) {
        set_error("exa_dg_riemann_corrector_cfl: the eigenvalue of this term set depends on position / time -- its CFL scan needs the node coordinates (exa_pde_eval_device_at)");
        return EXA_ERR_INVALID;
    }
    if (p) {
        int rc = use_device(p->device);
        if (rc) return rc;
        hipError_t e = hipMemsetAsync(lambda_dev, 0, sizeof(double), (hipStream_t)stream);
        if (e != hipSuccess) { set_error("exa_dg_riemann_corrector_cfl: memset: %s", hipGetErrorString(e)); return EXA_ERR_HIP; }
    }
    return riemann_corrector(p, u_dev, trace_dev, ghost_dev, lo, hi, dt, dx, lambda_dev, stream);
}

static int riemann_corrector(exa_dg_plan* p, double* u_dev, const double* trace_dev, const double* const* ghost_dev,
                             const long* lo, const long* hi, double dt, const double* dx, double* lambda_dev, void* stream) {
    if (!p || !u_dev || !trace_dev || !dx) { set_error("exa_dg_riemann_corrector: NULL argument"); return EXA_ERR_INVALID; }
    StageBBox box;
    box.lam = lambda_dev;
    CellBox cb;
    int rcb = make_box(p, lo, hi, &cb);
    if (rcb) return rcb;
    for (int d = 0; d < 3; d++) { box.nc[d] = cb.nc[d]; box.lo[d] = cb.lo[d]; box.nb[d] = cb.nb[d]; }
    for (int f = 0; f < 6; f++) box.ghost[f] = (ghost_dev && f < 2 * p->dim) ? ghost_dev[f] : nullptr;
    int rc = use_device(p->device);
    if (rc) return rc;
    double idx[3];
    inv_dx(p, dx, idx);
    return p->tab->stage_b(p->N, u_dev, trace_dev, &box, p->ncells, dt, idx, &p->ops, (hipStream_t)stream);
}

int exa_dg_plan_set_origin_time(exa_dg_plan* p, const double* origin, double t) {
    if (!p) { set_error("exa_dg_plan_set_origin_time: NULL plan"); return EXA_ERR_INVALID; }
    for (int d = 0; d < 3; d++) p->ops.origin[d] = (origin && d < p->dim) ? origin[d] : 0.0;
    p->ops.time = t;
    return EXA_OK;
}

int exa_dg_has_corrector_predictor(const exa_dg_plan* p) {
    return (p && p->tab->has_stage_ba) ? p->tab->has_stage_ba(p->N, p->n_it, p->ops.stage_a_variant) : 0;
}

int exa_dg_corrector_predictor(exa_dg_plan* p, double* u_dev, const double* trace_in_dev, double* trace_out_dev,
                               const double* const* ghost_dev, const long* lo, const long* hi, double dt_prev, double dt,
                               const double* dx, double* u_plain_dev, void* stream) {
    if (!p || !u_dev || !trace_in_dev || !trace_out_dev || !dx) { set_error("exa_dg_corrector_predictor: NULL argument"); return EXA_ERR_INVALID; }
    if (trace_in_dev == trace_out_dev) { set_error("exa_dg_corrector_predictor: the new traces need an array of their own"); return EXA_ERR_INVALID; }
    if (!exa_dg_has_corrector_predictor(p)) {
        set_error("exa_dg_corrector_predictor: not built for this plan (3-D, N = 6, register-resident stage A, n_picard >= 1)");
        return EXA_ERR_INVALID;
    }
    for (int d = 0; d < p->dim; d++)
        if (!(dx[d] > 0.0)) { set_error("dx[%d] must be > 0", d); return EXA_ERR_INVALID; }
    CellBox box;
    int rc = make_box(p, lo, hi, &box);
    if (rc) return rc;
    rc = use_device(p->device);
    if (rc) return rc;
    double idx[3];
    inv_dx(p, dx, idx);
    const double* gh[6];
    for (int f = 0; f < 6; f++) gh[f] = (ghost_dev && f < 2 * p->dim) ? ghost_dev[f] : nullptr;
    return p->tab->stage_ba(p->N, u_dev, trace_in_dev, trace_out_dev, p->ncells, &box, gh, dt_prev, dt, idx, p->n_it, &p->ops, u_plain_dev,
                            (hipStream_t)stream);
}

int exa_dg_pack_face(exa_dg_plan* p, const double* trace_dev, int d, int side, double* buf_dev, void* stream) {
    if (!p || !trace_dev || !buf_dev || d < 0 || d >= p->dim || side < 0 || side > 1) { set_error("exa_dg_pack_face: bad argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    // trace[(d*2+side)][cx][cy][cz][TS] -> the layer c_d = 0 (side 0) or nc_d-1 (side 1) as a strided 2-D copy
    const long TS = 2L * p->nv * lpow(p->N, p->dim - 1);
    long outer = 1, inner = 1;
    for (int a = 0; a < d; a++) outer *= p->nc[a];
    for (int a = d + 1; a < 3; a++) inner *= p->nc[a];
    const long layer = side ? p->nc[d] - 1 : 0;
    const double* src = trace_dev + ((long)(d * 2 + side) * p->ncells + layer * inner) * TS;
    EXA_HIP(hipMemcpy2DAsync(buf_dev, (size_t)inner * TS * sizeof(double), src, (size_t)p->nc[d] * inner * TS * sizeof(double),
                             (size_t)inner * TS * sizeof(double), (size_t)outer, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EXA_OK;
}

/* ---- FV subcell limiter glue ------------------------------------------------------ */

int exa_lim_operators(const exa_dg_plan* p, double* P, double* R) {
    if (!p || !P || !R) { set_error("exa_lim_operators: NULL argument"); return EXA_ERR_INVALID; }
    if (build_limiter_operators(&p->ops, 2 * p->N - 1, P, R) != 0) { set_error("limiter operator construction failed"); return EXA_ERR_INVALID; }
    return EXA_OK;
}

long exa_lim_patch_count(const exa_dg_plan* p) { return p ? lpow(2 * p->N + 1, p->dim) * p->nv : 0; }

static int lim_tables(exa_dg_plan* p) {
    if (p->ops.lim) return EXA_OK;
    const int N = p->N, Ns = 2 * N - 1;
    std::vector<double> t(2 * (size_t)N * Ns);
    if (build_limiter_operators(&p->ops, Ns, t.data(), t.data() + (size_t)N * Ns) != 0) { set_error("limiter operator construction failed"); return EXA_ERR_INVALID; }
    EXA_HIP(hipMalloc(&p->ops.lim, t.size() * sizeof(double)));
    EXA_HIP(hipMemcpy(p->ops.lim, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    return EXA_OK;
}

int exa_dg_project_patches_ghost(exa_dg_plan* p, const double* u_dev, const long* cells_dev, long n, double* patch_dev,
                                 const double* const* ghost_layers_dev, void* stream) {
    if (!p || !u_dev || (n > 0 && (!cells_dev || !patch_dev)) || n < 0) { set_error("exa_dg_project_patches: bad argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    rc = lim_tables(p);
    if (rc) return rc;
    const int Ns = 2 * p->N - 1;
    LimGhosts gh{};
    for (int f = 0; f < 6; f++) gh.layer[f] = (ghost_layers_dev && f < 2 * p->dim) ? ghost_layers_dev[f] : nullptr;
    return limiter_project(p->dim, p->N, Ns, p->nv, p->nc, u_dev, cells_dev, n, patch_dev, static_cast<const double*>(p->ops.lim), &gh,
                           (hipStream_t)stream);
}

int exa_dg_project_patches(exa_dg_plan* p, const double* u_dev, const long* cells_dev, long n, double* patch_dev, void* stream) {
    return exa_dg_project_patches_ghost(p, u_dev, cells_dev, n, patch_dev, nullptr, stream);
}

long exa_lim_face_layer_count(const exa_dg_plan* p, int d) {
    if (!p || d < 0 || d >= p->dim) return 0;
    long nt = 1;
    for (int b = 0; b < p->dim; b++)
        if (b != d) nt *= p->nc[b];
    return nt * lpow(2 * p->N - 1, p->dim - 1) * p->nv;
}

int exa_lim_face_layers(exa_dg_plan* p, const double* u_dev, int d, int side, const double* need_dev, double* out_dev, void* stream) {
    if (!p || !u_dev || !out_dev || d < 0 || d >= p->dim || side < 0 || side > 1) { set_error("exa_lim_face_layers: bad argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    rc = lim_tables(p);
    if (rc) return rc;
    return limiter_face_layers(p->dim, p->N, 2 * p->N - 1, p->nv, p->nc, u_dev, d, side, need_dev, out_dev,
                               static_cast<const double*>(p->ops.lim), (hipStream_t)stream);
}

int exa_dg_reconstruct_patches(exa_dg_plan* p, const double* patch_dev, const long* cells_dev, long n, double* u_dev, void* stream) {
    if (!p || !u_dev || (n > 0 && (!cells_dev || !patch_dev)) || n < 0) { set_error("exa_dg_reconstruct_patches: bad argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    rc = lim_tables(p);
    if (rc) return rc;
    const int Ns = 2 * p->N - 1;
    return limiter_reconstruct(p->dim, p->N, Ns, p->nv, patch_dev, cells_dev, n, u_dev,
                               static_cast<const double*>(p->ops.lim) + (size_t)p->N * Ns, (hipStream_t)stream);
}

int exa_dg_max_eigenvalue(exa_dg_plan* p, const double* u_dev, double* lambda_dev, void* stream) {
    if (!p || !u_dev || !lambda_dev) { set_error("exa_dg_max_eigenvalue: NULL argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    return p->tab->maxeig(u_dev, p->ncells * lpow(p->N, p->dim), lambda_dev, (hipStream_t)stream);
}

int exa_dg_has_fused_step(const exa_dg_plan* p) { return (p && p->n_it == 0 && p->tab->fused_single) ? 1 : 0; }

int exa_dg_step_fused(exa_dg_plan* p, const double* u_in_dev, double* u_out_dev, double dt, const double* dx, void* stream) {
    if (!p || !u_in_dev || !u_out_dev || !dx || u_in_dev == u_out_dev) { set_error("exa_dg_step_fused: bad argument (input and output must differ)"); return EXA_ERR_INVALID; }
    if (!exa_dg_has_fused_step(p)) { set_error("exa_dg_step_fused: only for the single-stage scheme (n_picard = 0) in 2-D"); return EXA_ERR_INVALID; }
    for (int d = 0; d < p->dim; d++)
        if (!(dx[d] > 0.0)) { set_error("dx[%d] must be > 0", d); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    double idx[3];
    inv_dx(p, dx, idx);
    return p->tab->fused_single(p->N, u_in_dev, u_out_dev, p->nc, dt, idx, &p->ops, (hipStream_t)stream) == 0 ? EXA_OK : EXA_ERR_HIP;
}

int exa_dg_step_periodic(exa_dg_plan* p, double* u_dev, double* trace_dev, double dt, const double* dx, int n_steps,
                         void* stream) {
    for (int s = 0; s < n_steps; s++) {
        int rc = exa_dg_predictor_volume(p, u_dev, trace_dev, dt, dx, stream);
        if (rc) return rc;
        rc = exa_dg_riemann_corrector(p, u_dev, trace_dev, nullptr, nullptr, nullptr, dt, dx, stream);
        if (rc) return rc;
    }
    return EXA_OK;
}

int exa_dg_step_host(exa_dg_plan* p, double* u_host, double dt, const double* dx, int n_steps) {
    if (!p || !u_host || !dx) { set_error("exa_dg_step_host: NULL argument"); return EXA_ERR_INVALID; }
    int rc = use_device(p->device);
    if (rc) return rc;
    double *u = nullptr, *tr = nullptr;
    const size_t ub = (size_t)exa_dg_dof_count(p) * sizeof(double), tb = (size_t)exa_dg_trace_count(p) * sizeof(double);
    EXA_HIP(hipMalloc(&u, ub));
    hipError_t e = hipMalloc(&tr, tb);
    if (e == hipSuccess) e = hipMemcpy(u, u_host, ub, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = exa_dg_step_periodic(p, u, tr, dt, dx, n_steps, nullptr);
        if (rc == EXA_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(u_host, u, ub, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(u);
    (void)hipFree(tr);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("exa_dg_step_host: %s", hipGetErrorString(e)); return EXA_ERR_HIP; }
    return EXA_OK;
}

}  // extern "C"
