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
#include "exa_launch.hpp"
#include "exa_pde.hpp"
#ifdef EXA_USER_PDE_HEADER
#include EXA_USER_PDE_HEADER      // struct exa::UserPDE (exahype_amd/pde_codegen.py)
#endif

namespace exa {

constexpr int MAXV = 8;
typedef double v2d __attribute__((ext_vector_type(2)));

// STAGE = true (small patches): the workgroup's `ppb` patches are one contiguous block of HBM; it is
// copied into LDS with coalesced 16-byte loads and the stencil reads LDS (the AoS stencil reads straight
// from HBM touch ~40 cache lines per wave instruction: 1.7 TB/s; staged: see DESIGN.md 4.3).
// SHAPE: compile-time (patch_size, halo_size, n_real, n_real + n_aux), or all zero for run-time values.  The reference's
// own configuration (4, 1, 5, 10) is built specialised: the variable masks and the 64-bit index arithmetic fold away
// (a wave executed ~1 030 VALU instructions in the generic build; the arithmetic and its order are the same).
template <int TP, int TH, int TM, int TV> struct FvShape {
    static constexpr int P = TP, H = TH, M = TM, V = TV;
};
using FvRuntimeShape = FvShape<0, 0, 0, 0>;

template <int DIM, class PDE, int MODE, int CPT, int NT, bool STAGE, class SHAPE = FvRuntimeShape>
__global__ void __launch_bounds__(NT)
fv_rusanov_kernel(double* __restrict__ Q, int P_rt, int H_rt, int m_rt, int V_rt, double dt, double dt_over_h, long n_patches,
                  int ppb, const long* __restrict__ slot) {
    const int P = SHAPE::P ? SHAPE::P : P_rt, H = SHAPE::P ? SHAPE::H : H_rt;
    const int m = SHAPE::P ? SHAPE::M : m_rt, V = SHAPE::P ? SHAPE::V : V_rt;
    extern __shared__ __attribute__((aligned(16))) double fv_lds[];
    // `ppb` small patches share one workgroup (the reference's 4x4 patch has 16 volumes: one patch per
    // 256-thread workgroup would idle 94 % of the lanes); large patches use ppb = 1 and CPT volumes per thread.
    const int S = P + 2 * H;
    const long vol = (DIM == 3) ? (long)S * S * S : (long)S * S;
    const int ncell = (DIM == 3) ? P * P * P : P * P;
    const int pl = (CPT == 1) ? (int)threadIdx.x / ncell : 0;            // patch slot of this thread
    const long patch = (long)blockIdx.x * ppb + pl;
    // slot (optional): one entry per patch, < 0 = this patch is not in use (exa_fv_time_step_device_masked: the
    // number of patches in use is known on the device only, the launch covers the array's capacity)
    const bool live = pl < ppb && patch < n_patches && (!slot || slot[patch] >= 0);
    double* Qg = Q + (live ? patch : 0) * vol * V;              // this thread's patch in HBM (writes)
    const double* Qp = Qg;                                       // ... and where the stencil reads it
    if constexpr (STAGE) {
        const long first = (long)blockIdx.x * ppb;
        const long npatch = (n_patches - first < ppb) ? n_patches - first : ppb;
        const long total = npatch * vol * V;                     // doubles in this workgroup's block
        const double* src = Q + first * vol * V;
        const bool al16 = ((reinterpret_cast<unsigned long long>(src) & 15) == 0);
        if (al16) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(fv_lds);
            for (long i = threadIdx.x; i < total / 2; i += NT) d2[i] = s2[i];
            if ((total & 1) && threadIdx.x == 0) fv_lds[total - 1] = src[total - 1];
        } else {
            for (long i = threadIdx.x; i < total; i += NT) fv_lds[i] = src[i];
        }
        __syncthreads();
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
        int co[3];
        if constexpr (DIM == 3) { co[0] = id / (P * P) + H; co[1] = (id / P) % P + H; co[2] = id % P + H; }
        else { co[0] = id / P + H; co[1] = id % P + H; co[2] = 0; }
        const long c = co[0] * st[0] + co[1] * st[1] + co[2] * st[2];
        cidx[k] = c;
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
#pragma unroll
                for (int v = 0; v < MAXV; v++) {
                    qP[v] = v < m ? Qp[(c + st[d]) * V + v] : 0.0;
                    qM[v] = v < m ? Qp[(c - st[d]) * V + v] : 0.0;
                }
                if (co[d] + 1 < P + H) PDE::flux_rt(qP, d, Fp);
                if (co[d] - 1 >= H) PDE::flux_rt(qM, d, Fm);
#pragma unroll
                for (int v = 0; v < MAXV; v++) acc[v] = acc[v] - 0.5 * Fp[v] + 0.5 * Fm[v];
            }
            // L8/L9 (test.cpp:78-95): variable 0 only, reads the ORIGINAL Q
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double qP[MAXV], qM[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) {
                    qP[v] = v < m ? Qp[(c + st[d]) * V + v] : 0.0;
                    qM[v] = v < m ? Qp[(c - st[d]) * V + v] : 0.0;
                }
                const double lc = PDE::maxeig(qc, d);
                const double lp = (co[d] + 1 < P + H) ? PDE::maxeig(qP, d) : 0.0;
                const double lm = (co[d] - 1 >= H) ? PDE::maxeig(qM, d) : 0.0;
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
#pragma unroll
                for (int v = 0; v < MAXV; v++) {
                    qpp[v] = v < m ? Qp[(c + st[d]) * V + v] : 0.0;
                    qmp[v] = v < m ? Qp[(c - st[d]) * V + v] : 0.0;
                }
                const double* qcp = qc;
                const double lc = PDE::maxeig(qcp, d);
                const double sp = fmax(lc, PDE::maxeig(qpp, d));
                const double sm = fmax(PDE::maxeig(qmp, d), lc);
                double Fc[MAXV], Fn[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) { Fc[v] = 0.0; Fn[v] = 0.0; }
                PDE::flux_rt(qcp, d, Fc);
                PDE::flux_rt(qpp, d, Fn);
#pragma unroll
                for (int v = 0; v < MAXV; v++)
                    if (v < m) acc[v] += 0.5 * (Fc[v] + Fn[v]) - 0.5 * sp * (qpp[v] - qc[v]);
#pragma unroll
                for (int v = 0; v < MAXV; v++) Fn[v] = 0.0;
                PDE::flux_rt(qmp, d, Fn);
#pragma unroll
                for (int v = 0; v < MAXV; v++)
                    if (v < m) acc[v] -= 0.5 * (Fn[v] + Fc[v]) - 0.5 * sm * (qc[v] - qmp[v]);
            }
#pragma unroll
            for (int v = 0; v < MAXV; v++) nv[k][v] = qc[v] - dt_over_h * acc[v];
        }
    }
    // every read of this patch is done (loads feed the values above) before any write
    __syncthreads();
    bool rows = false;
    if constexpr (STAGE) {
        // auxiliary variables present (V even, rows 16-byte aligned): results go into the LDS copy first, then
        // whole interior rows (P*V contiguous doubles) stream back with coalesced 16-byte stores
        rows = ((reinterpret_cast<unsigned long long>(Q) & 15) == 0) && (V % 2 == 0) && (V > m);
    }
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
        const long first = (long)blockIdx.x * ppb;
        const long npatch = (n_patches - first < ppb) ? n_patches - first : ppb;
        const int rows_pp = (DIM == 3) ? P * P : P;                  // interior rows per patch
        const int r2 = P * V / 2;                                    // double2 per row
        double* dst = Q + first * vol * V;
        for (long e = threadIdx.x; e < npatch * rows_pp * r2; e += NT) {
            const long row = e / r2;
            const int x = (int)(e - row * r2);
            const long pp = row / rows_pp;
            const int rr = (int)(row - pp * rows_pp);
            const long o = pp * vol * V + ((DIM == 3) ? ((long)(rr / P + H) * S * S + (long)(rr % P + H) * S + H) : ((long)(rr + H) * S + H)) * V;
            // streaming store: the rows start at 80-byte offsets, i.e. partial cache lines -- do not allocate them in L2
            __builtin_nontemporal_store(reinterpret_cast<const v2d*>(fv_lds + o)[x], reinterpret_cast<v2d*>(dst + o) + x);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            if (cidx[k] < 0) continue;
#pragma unroll
            for (int v = 0; v < MAXV; v++)
                if (v < m) Qg[cidx[k] * V + v] = nv[k][v];
        }
    }
}

// Large 3-D patches (cfg 4's limiter patch: 17^3 x 5 doubles = 196 KB with halo, more than LDS): the patch streams
// plane by plane (axis 0) through a 4-slot LDS ring, filled with coalesced 16-byte loads one plane ahead; thread
// (j, k) updates the volume (i, j, k) of the current plane from the ring.  The ring holds the OLD values, so the new
// values of plane i go straight back to HBM in place: every plane is read once and its interior written once.
template <class PDE, int MODE>
__global__ void __launch_bounds__(256)
fv_rusanov_slab_kernel(double* __restrict__ Q, int P, int H, int m, int V, double dt, double dt_over_h,
                       const long* __restrict__ slot) {
    extern __shared__ __attribute__((aligned(16))) double ring[];
    if (slot && slot[blockIdx.x] < 0) return;                      // patch not in use (workgroup-uniform)
    const int S = P + 2 * H;
    const int plane = S * S * V;                                  // doubles per plane
    double* Qp = Q + (long)blockIdx.x * S * plane;
    const int tid = threadIdx.x;
    const int j = tid / P + H, k = tid % P + H;
    const bool cell_ok = tid < P * P;
    auto load_plane = [&](int i) {
        const double* src = Qp + (long)i * plane;
        double* dst = ring + (i & 3) * plane;
        if (((reinterpret_cast<unsigned long long>(src) & 15) == 0) && (plane % 2 == 0)) {
            for (int x = tid; x < plane / 2; x += 256) reinterpret_cast<double2*>(dst)[x] = reinterpret_cast<const double2*>(src)[x];
        } else {
            for (int x = tid; x < plane; x += 256) dst[x] = src[x];
        }
    };
    load_plane(H - 1);
    load_plane(H);
    for (int i = H; i < P + H; i++) {
        load_plane(i + 1);                                         // slot (i+1)&3 was last read in iteration i-3
        __syncthreads();
        if (cell_ok) {
            const double* c0 = ring + (i & 3) * plane + (j * S + k) * V;
            const double* nb[3][2] = {{ring + ((i - 1) & 3) * plane + (j * S + k) * V, ring + ((i + 1) & 3) * plane + (j * S + k) * V},
                                      {c0 - S * V, c0 + S * V}, {c0 - V, c0 + V}};
            const int co[3] = {i, j, k};
            double qc[MAXV], out[MAXV];
#pragma unroll
            for (int v = 0; v < MAXV; v++) qc[v] = v < m ? c0[v] : 0.0;
            if constexpr (MODE == 0) {
                double acc[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) acc[v] = qc[v];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    double Fp[MAXV], Fm[MAXV], qP[MAXV], qM[MAXV];
#pragma unroll
                    for (int v = 0; v < MAXV; v++) { Fp[v] = 0.0; Fm[v] = 0.0; qP[v] = v < m ? nb[d][1][v] : 0.0; qM[v] = v < m ? nb[d][0][v] : 0.0; }
                    if (co[d] + 1 < P + H) PDE::flux_rt(qP, d, Fp);
                    if (co[d] - 1 >= H) PDE::flux_rt(qM, d, Fm);
#pragma unroll
                    for (int v = 0; v < MAXV; v++) acc[v] = acc[v] - 0.5 * Fp[v] + 0.5 * Fm[v];
                }
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    double qP[MAXV], qM[MAXV];
#pragma unroll
                    for (int v = 0; v < MAXV; v++) { qP[v] = v < m ? nb[d][1][v] : 0.0; qM[v] = v < m ? nb[d][0][v] : 0.0; }
                    const double lc = PDE::maxeig(qc, d);
                    const double lp = (co[d] + 1 < P + H) ? PDE::maxeig(qP, d) : 0.0;
                    const double lm = (co[d] - 1 >= H) ? PDE::maxeig(qM, d) : 0.0;
                    const double mp = lp > lc ? lp : lc;
                    const double mm = lm > lc ? lm : lc;
                    acc[0] = 0.5 * dt * ((-qP[0] + qc[0]) * mp + (qM[0] - qc[0]) * mm) + acc[0];
                }
#pragma unroll
                for (int v = 0; v < MAXV; v++) out[v] = acc[v];
            } else {
                double acc[MAXV];
#pragma unroll
                for (int v = 0; v < MAXV; v++) acc[v] = 0.0;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    double qpp[MAXV], qmp[MAXV], Fc[MAXV], Fn[MAXV];
#pragma unroll
                    for (int v = 0; v < MAXV; v++) { qpp[v] = v < m ? nb[d][1][v] : 0.0; qmp[v] = v < m ? nb[d][0][v] : 0.0; Fc[v] = 0.0; Fn[v] = 0.0; }
                    const double lc = PDE::maxeig(qc, d);
                    const double sp = fmax(lc, PDE::maxeig(qpp, d));
                    const double sm = fmax(PDE::maxeig(qmp, d), lc);
                    PDE::flux_rt(qc, d, Fc);
                    PDE::flux_rt(qpp, d, Fn);
#pragma unroll
                    for (int v = 0; v < MAXV; v++)
                        if (v < m) acc[v] += 0.5 * (Fc[v] + Fn[v]) - 0.5 * sp * (qpp[v] - qc[v]);
#pragma unroll
                    for (int v = 0; v < MAXV; v++) Fn[v] = 0.0;
                    PDE::flux_rt(qmp, d, Fn);
#pragma unroll
                    for (int v = 0; v < MAXV; v++)
                        if (v < m) acc[v] -= 0.5 * (Fn[v] + Fc[v]) - 0.5 * sm * (qc[v] - qmp[v]);
                }
#pragma unroll
                for (int v = 0; v < MAXV; v++) out[v] = qc[v] - dt_over_h * acc[v];
            }
            double* dst = Qp + (long)i * plane + (j * S + k) * V;
#pragma unroll
            for (int v = 0; v < MAXV; v++)
                if (v < m) dst[v] = out[v];
        }
        __syncthreads();                                           // plane i-1's slot may be refilled next iteration
    }
}

template <class PDE>
__global__ void pde_eval_kernel(int normal, long n, int stride, const double* __restrict__ Q, double* __restrict__ F,
                                double* __restrict__ lam) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (F) {
        double f[MAXV];
#pragma unroll
        for (int v = 0; v < MAXV; v++) f[v] = 0.0;
        PDE::flux_rt(&Q[i * stride], normal, f);
        for (int v = 0; v < PDE::NFLUX && v < stride; v++) F[i * stride + v] = f[v];
    }
    if (lam) lam[i] = PDE::maxeig(&Q[i * stride], normal);
}

template <int DIM, class PDE, int MODE>
static int fv_dispatch(int P, int H, int m, int V, long n_patches, double* Q, double dt, double h, const long* slot, hipStream_t s) {
    const long ncell = (DIM == 3) ? (long)P * P * P : (long)P * P;
    const double doh = (MODE == 1) ? dt / h : 0.0;
    const int S = P + 2 * H;
    const long pvol = (DIM == 3) ? (long)S * S * S : (long)S * S;
    if (ncell <= 256) {
        const int ppb = (int)(256 / ncell);
        const dim3 grid((unsigned)((n_patches + ppb - 1) / ppb));
        const size_t lds = (size_t)ppb * pvol * V * sizeof(double);
        if (DIM == 2 && P == 4 && H == 1 && m == 5 && V == 10)      // the reference's configuration (Batched_stateless.py:9)
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, true, FvShape<4, 1, 5, 10>>), grid, dim3(256), lds, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot);
        else if (lds <= 64 * 1024)       // staged: up to two workgroups per CU keep HBM requests in flight
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, true>), grid, dim3(256), lds, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot);
        else
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 256, false>), grid, dim3(256), 0, s, Q, P, H, m, V, dt, doh, n_patches, ppb, slot);
    } else if (ncell <= 1024) {
        const size_t lds = (size_t)pvol * V * sizeof(double);
        if (lds <= 64 * 1024)
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 1024, true>), dim3((unsigned)n_patches), dim3(1024), lds, s, Q, P, H, m, V, dt, doh, n_patches, 1, slot);
        else
            hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 1, 1024, false>), dim3((unsigned)n_patches), dim3(1024), 0, s, Q, P, H, m, V, dt, doh, n_patches, 1, slot);
    } else if (DIM == 3 && P * P <= 256 && (size_t)4 * S * S * V * sizeof(double) <= 64 * 1024) {
        // plane-streaming variant: 4-plane LDS ring, one workgroup per patch
        hipLaunchKernelGGL((fv_rusanov_slab_kernel<PDE, MODE>), dim3((unsigned)n_patches), dim3(256), (size_t)4 * S * S * V * sizeof(double), s,
                           Q, P, H, m, V, dt, doh, slot);
    } else if (ncell <= 4096) {
        hipLaunchKernelGGL((fv_rusanov_kernel<DIM, PDE, MODE, 4, 1024, false>), dim3((unsigned)n_patches), dim3(1024), 0, s, Q, P, H, m, V, dt, doh, n_patches, 1, slot);
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
static int fv_mode(int mode, int P, int H, int m, int V, long n_patches, double* Q, double dt, double h, const long* slot, hipStream_t s) {
    if (mode == 0) return fv_dispatch<DIM, PDE, 0>(P, H, m, V, n_patches, Q, dt, h, slot, s);
    return fv_dispatch<DIM, PDE, 1>(P, H, m, V, n_patches, Q, dt, h, slot, s);
}

#ifdef EXA_USER_PDE_HEADER
}  // namespace exa
// user-PDE side library: the same fused kernel instantiated for exa::UserPDE
extern "C" int exa_user_nv() { return exa::UserPDE::NV; }
extern "C" int exa_user_fv_launch(int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, double* Q, double dt,
                                  double h, const long* slot, void* stream) {
    using namespace exa;
    const int V = n_real + n_aux;
    if (n_real > MAXV || n_real < UserPDE::NV) { set_error("user PDE evolves %d variables; n_real = %d", UserPDE::NV, n_real); return -1; }
    if (n_patches <= 0) return 0;
    if (dim == 2) return fv_mode<2, UserPDE>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, (hipStream_t)stream);
    if (dim == 3 && UserPDE::MAXDIM >= 3) return fv_mode<3, UserPDE>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, (hipStream_t)stream);
    set_error("user PDE: no FV kernel for dim %d", dim);
    return -1;
}
extern "C" int exa_user_pde_eval(int normal, long n, int stride, const double* Q, double* F, double* lam, void* stream) {
    using namespace exa;
    if (n <= 0) return 0;
    hipLaunchKernelGGL((pde_eval_kernel<UserPDE>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, normal, n, stride, Q, F, lam);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("pde_eval launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
namespace exa {
#else
int fv_launch(int mode, int dim, int P, int H, int n_real, int n_aux, long n_patches, int pde, double* Q, double dt,
              double h, const long* slot, hipStream_t s) {
    const int V = n_real + n_aux;
    if (n_real > MAXV) { set_error("n_real = %d exceeds %d", n_real, MAXV); return -1; }
    if (n_patches <= 0) return 0;
    if (pde >= 100) return user_fv_launch(pde, mode, dim, P, H, n_real, n_aux, n_patches, Q, dt, h, slot, s);
    if (dim == 2) {
        if (pde == 0) return fv_mode<2, EulerRef2D>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s);
        if (pde == 1) return fv_mode<2, Euler>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s);
        if (pde == 2) return fv_mode<2, Advection<MAXV>>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s);
    } else if (dim == 3) {
        if (pde == 1) return fv_mode<3, Euler>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s);
        if (pde == 2) return fv_mode<3, Advection<MAXV>>(mode, P, H, n_real, V, n_patches, Q, dt, h, slot, s);
    }
    set_error("FV Rusanov: no kernel for dim %d, pde %d", dim, pde);
    return -1;
}

int pde_eval_launch(int pde, int normal, long n, int stride, const double* Q, double* F, double* lam, hipStream_t s) {
    if (n <= 0) return 0;
    if (pde >= 100) return user_pde_eval(pde, normal, n, stride, Q, F, lam, s);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (pde == 0) hipLaunchKernelGGL((pde_eval_kernel<EulerRef2D>), grid, dim3(256), 0, s, normal, n, stride, Q, F, lam);
    else if (pde == 1) hipLaunchKernelGGL((pde_eval_kernel<Euler>), grid, dim3(256), 0, s, normal, n, stride, Q, F, lam);
    else if (pde == 2) hipLaunchKernelGGL((pde_eval_kernel<Advection<1>>), grid, dim3(256), 0, s, normal, n, stride, Q, F, lam);
    else { set_error("unknown pde %d", pde); return -1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("pde_eval launch: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

#endif

}  // namespace exa
