// Device twins of the reference's user PDE terms (`Unit test/Functions.cpp:9-66`).
//
// The reference resolves `Flux` / `maxEigenvalue` / `max` at link time to host
// C++ (`Unit test/Functions.h:2-4`); a HIP kernel cannot call those, so the
// kernels are templated on one of these structs instead.
//
// Interface every PDE struct provides:
//   NV        variables the scheme evolves (n_real)
//   NFLUX     leading entries of F that flux() writes (rest are left alone)
//   NAUX      per-node cached scalars shared by the flux in every direction (0 allowed: no LDS slot, arrays sized nz(NAUX))
//   aux(q, a)                  a[NAUX] from q[NV]   (Euler: 1/rho and p)
//   flux<d>(q, a, F)           F[NV] for normal d using the cached scalars
//   flux_scaled<d>(q, a, s, F) s * F[NV] with the scale folded into the two cached scalars
//   flux_rt(q, d, F)           same, run-time normal, no cache (FV path, traces)
//   maxeig(q, d)               largest absolute eigenvalue along d
//   maxeig_fast(q, d)          same to rounding with the fast reciprocal / square root (ADER-DG kernels)
//
// Arithmetic order follows Functions.cpp exactly (irho first, p from irho, coeff =
// irho*Q[normal+1], F[normal+1] += p), so results agree with the CPU path to
// rounding; the FV faithful kernel compiles these with FP contraction off and is
// bit-exact.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace exa {

// maximum for the CFL reductions that KEEPS a NaN (fmax drops it: a state that has blown up -- negative pressure, NaN eigenvalue -- would yield a finite
// lambda_max and the time loop would go on silently; the NaN's bit pattern also wins the unsigned atomicMax the reductions end in)
__device__ inline double nan_max(double a, double b) { return (a != a) ? a : ((b != b) ? b : fmax(a, b)); }


constexpr double GAMMA = 1.4;

// array extent for NAUX cached scalars: a term set may have none (zero-length arrays are not permitted in device code)
__host__ __device__ constexpr int nz(int n) { return n > 0 ? n : 1; }

// Optional member of a PDE struct: `static constexpr bool HAS_SOURCE = true` with `source(q, S)` = the algebraic source S(q)[NV] of
// q_t + div F(q) = S(q) (the hook the reference's harness declares next to flux and maxEigenvalue, `Unit test/correctness_test.cpp:16-23`;
// SURVEY.md 8(f)-2).  The built-in term sets have none; pde_codegen.SympyPDE(source=...) generates one.
template <class P, class = void> struct pde_has_source : std::false_type {};
template <class P> struct pde_has_source<P, std::void_t<decltype(P::HAS_SOURCE)>> : std::bool_constant<P::HAS_SOURCE> {};

// Optional members of a PDE struct: `static constexpr bool HAS_XT = true` with flux_xt(q, x, t, d, F), maxeig_xt(q, x, t, d) and (with
// HAS_SOURCE) source_xt(q, x, t, S) -- the terms may depend on the position x[3] (FV: the volume centre; ADER-DG: the node) and the time t, as the hooks of the reference's
// harness are declared (`Unit test/correctness_test.cpp:16-41`: flux / maxEigenvalue / sourceTerm(Q, x, h, t, dt, ...)).  The built-in
// term sets have none, and for them the coordinates below are dead code: their kernels are unchanged.
template <class P, class = void> struct pde_has_xt : std::false_type {};
template <class P> struct pde_has_xt<P, std::void_t<decltype(P::HAS_XT)>> : std::bool_constant<P::HAS_XT> {};
// ... and whether the FLUX is among them (a generated term set says FLUX_XT = false when only its source / eigenvalue see x, t: the ADER-DG kernels then
// use its tuned flux members and hand the coordinates to the source alone)
template <class P, class = void> struct pde_flux_xt : pde_has_xt<P> {};
template <class P> struct pde_flux_xt<P, std::void_t<decltype(P::FLUX_XT)>> : std::bool_constant<pde_has_xt<P>::value && P::FLUX_XT> {};
template <class PDE> __device__ inline void fv_flux(const double* q, const double* x, double t, int d, double* F) {
    if constexpr (pde_has_xt<PDE>::value) PDE::flux_xt(q, x, t, d, F);
    else PDE::flux_rt(q, d, F);
}
template <class PDE> __device__ inline double fv_eig(const double* q, const double* x, double t, int d) {
    if constexpr (pde_has_xt<PDE>::value) return PDE::maxeig_xt(q, x, t, d);
    else return PDE::maxeig(q, d);
}
template <class PDE> __device__ inline void fv_source(const double* q, const double* x, double t, double* S) {
    if constexpr (pde_has_xt<PDE>::value) PDE::source_xt(q, x, t, S);
    else PDE::source(q, S);
}
// Optional: `static constexpr bool HAS_NCP = true` with ncp(q, dq, d, out) (HAS_XT: ncp_xt(q, dq, x, t, d, out)) = B_d(q) dq, the
// non-conservative product of q_t + div F(q) + B(q) . grad q = S(q) -- the `ncp` slot of the kernel the reference's harness targets
// (`Unit test/correctness_test.cpp:145-155`: <Flux, ncp, Source, Eigen>).  Corrected Rusanov mode only: across a face the jump term
// D = B_d((q_L + q_R) / 2) (q_R - q_L) (straight path, midpoint) goes half to either side, as ExaHyPE 2's FV Rusanov solver does.
template <class P, class = void> struct pde_has_ncp : std::false_type {};
template <class P> struct pde_has_ncp<P, std::void_t<decltype(P::HAS_NCP)>> : std::bool_constant<P::HAS_NCP> {};
template <class PDE> __device__ inline void fv_ncp(const double* q, const double* dq, const double* x, double t, int d, double* out) {
    if constexpr (pde_has_xt<PDE>::value) PDE::ncp_xt(q, dq, x, t, d, out);
    else PDE::ncp(q, dq, d, out);
}

// The same terms for the ADER-DG kernels (tolerance 1e-10): generated term sets carry `_fast` twins whose reciprocals / square roots use the
// fast sequences below (pde_codegen.py); a term set without them is evaluated as it is.
template <class P, class = void> struct pde_has_fast_xt : std::false_type {};
template <class P> struct pde_has_fast_xt<P, std::void_t<decltype(&P::flux_xt_fast)>> : std::true_type {};
template <class P, class = void> struct pde_has_fast_ncp : std::false_type {};
template <class P> struct pde_has_fast_ncp<P, std::void_t<decltype(&P::ncp_fast)>> : std::true_type {};
template <class P, class = void> struct pde_has_fast_ncp_xt : std::false_type {};
template <class P> struct pde_has_fast_ncp_xt<P, std::void_t<decltype(&P::ncp_xt_fast)>> : std::true_type {};
template <class PDE> __device__ inline void dg_flux_xt(const double* q, const double* x, double t, int d, double* F) {
    if constexpr (pde_has_fast_xt<PDE>::value) PDE::flux_xt_fast(q, x, t, d, F);
    else PDE::flux_xt(q, x, t, d, F);
}
template <class PDE> __device__ inline void dg_ncp(const double* q, const double* dq, const double* x, double t, int d, double* out) {
    if constexpr (pde_has_xt<PDE>::value) {
        if constexpr (pde_has_fast_ncp_xt<PDE>::value) PDE::ncp_xt_fast(q, dq, x, t, d, out);
        else PDE::ncp_xt(q, dq, x, t, d, out);
    } else {
        if constexpr (pde_has_fast_ncp<PDE>::value) PDE::ncp_fast(q, dq, d, out);
        else PDE::ncp(q, dq, d, out);
    }
}

// 1/x from v_rcp_f64 + EXA_RCP_NR Newton steps.  Measured on MI355X against the IEEE quotient
// (scripts/rcp_accuracy.hip, 4M values): bare v_rcp_f64 2.6e8 ulp, one step <= 11 ulp (2.5e-15 relative), two
// steps exact.  One step is the default: used by the ADER-DG kernels only (tolerance 1e-10; 2 % of stage A);
// the FV faithful kernel keeps the correctly rounded division the reference's CPU build performs.
#ifndef EXA_RCP_NR
#define EXA_RCP_NR 1
#endif
__device__ inline double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
#if EXA_RCP_NR >= 2
    r = fma(fma(-x, r, 1.0), r, r);
#endif
    return r;
}

// sqrt(x) for x >= 0 from v_rsq_f64 + two Goldschmidt steps (<= 1 ulp), a quarter of the IEEE sequence
__device__ inline double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    return x > 0.0 ? g : 0.0;
}

// Functions.cpp:9-62 as the reference compiles it (`Dimensions` undefined -> 2-D
// branch): state (rho, rho u, rho v, E) in Q[0..3]; F[0..3] written.
struct EulerRef2D {
    static constexpr int NV = 5;      // the reference runs it with n_real = 5 (F[4] never written)
    static constexpr int NFLUX = 4;
    static constexpr int NAUX = 2;
    static constexpr int MAXDIM = 2;
    __device__ static inline void aux(const double* q, double* a) {
        const double irho = 1.0 / q[0];
        a[0] = irho;
        a[1] = (GAMMA - 1) * (q[3] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2]));
    }
    __device__ static inline void aux_fast(const double* q, double* a) {
        const double irho = fast_rcp(q[0]);
        a[0] = irho;
        a[1] = (GAMMA - 1) * (q[3] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2]));
    }
    template <int D> __device__ static inline void flux(const double* q, const double* a, double* F) {
        const double coeff = a[0] * q[D + 1];
        F[0] = coeff * q[0];
        F[1] = coeff * q[1];
        F[2] = coeff * q[2];
        F[3] = coeff * q[3] + coeff * a[1];
        F[D + 1] += a[1];
        F[4] = 0.0;
    }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double* a, double sc, double* F) {
        const double coeff = a[0] * q[D + 1] * sc;
        F[0] = coeff * q[0];
        F[1] = coeff * q[1];
        F[2] = coeff * q[2];
        F[3] = coeff * (q[3] + a[1]);
        F[D + 1] += a[1] * sc;
        F[4] = 0.0;
    }
    __device__ static inline void flux_rt(const double* q, int d, double* F) {
        const double irho = 1.0 / q[0];
        const double p = (GAMMA - 1) * (q[3] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2]));
        const double coeff = irho * q[d + 1];
        F[0] = coeff * q[0];
        F[1] = coeff * q[1];
        F[2] = coeff * q[2];
        F[3] = coeff * q[3] + coeff * p;
        F[d + 1] += p;
    }
    __device__ static inline double maxeig(const double* q, int d) {
        const double irho = 1.0 / fabs(q[0]);
        const double p = (GAMMA - 1) * (q[3] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2]));
        const double c = sqrt(GAMMA * fabs(p) * irho);
        const double un = q[d + 1] * irho;
        return fmax(fabs(un - c), fabs(un + c));
    }
    // same value to rounding (fast reciprocal and square root, no dynamic register index); ADER-DG kernels only
    __device__ static inline double maxeig_fast(const double* q, int d) {
        const double irho = fast_rcp(fabs(q[0]));
        const double p = (GAMMA - 1) * (q[3] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2]));
        const double c = fast_sqrt(GAMMA * fabs(p) * irho);
        const double un = (d == 0 ? q[1] : q[2]) * irho;
        return fmax(fabs(un - c), fabs(un + c));
    }
};

// Same arithmetic with a 3-component momentum: (rho, m0, m1, m2, E).  This is the
// 3-D branch of Functions.cpp without its stray `F[3] = ...` overwrite
// (SURVEY.md Appendix B-4).
struct Euler {
    static constexpr int NV = 5;
    static constexpr int NFLUX = 5;
    static constexpr int NAUX = 2;
    static constexpr int MAXDIM = 3;
    __device__ static inline void aux(const double* q, double* a) {
        const double irho = 1.0 / q[0];
        a[0] = irho;
        a[1] = (GAMMA - 1) * (q[4] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
    }
    __device__ static inline void aux_fast(const double* q, double* a) {
        const double irho = fast_rcp(q[0]);
        a[0] = irho;
        a[1] = (GAMMA - 1) * (q[4] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
    }
    template <int D> __device__ static inline void flux(const double* q, const double* a, double* F) {
        const double coeff = a[0] * q[D + 1];
        F[0] = coeff * q[0];
        F[1] = coeff * q[1];
        F[2] = coeff * q[2];
        F[3] = coeff * q[3];
        F[4] = coeff * q[4] + coeff * a[1];
        F[D + 1] += a[1];
    }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double* a, double sc, double* F) {
        const double coeff = a[0] * q[D + 1] * sc;
        F[0] = coeff * q[0];
        F[1] = coeff * q[1];
        F[2] = coeff * q[2];
        F[3] = coeff * q[3];
        F[4] = coeff * (q[4] + a[1]);
        F[D + 1] += a[1] * sc;
    }
    // Normal chosen per LANE (exa_dg_reg.hpp: the lanes of a wave hold pencils of different directions): per-lane masks psc[k] =
    // scale for the lane's own direction, 0 for the others, pick the normal momentum and place the pressure -- straight-line code,
    // no select (a `d == 0 ? q[1] : ...` chain is turned into a run-time-indexed stack array, i.e. scratch traffic).
    struct Dir {
        double psc[3];
    };
    __device__ static inline void dir_init(Dir& c, int d, double sc) {
        c.psc[0] = d == 0 ? sc : 0.0;
        c.psc[1] = d == 1 ? sc : 0.0;
        c.psc[2] = d == 2 ? sc : 0.0;
    }
    __device__ static inline void flux_scaled_dir(const double* q, const double* a, const Dir& c, double* F) {
        const double coeff = a[0] * fma(q[3], c.psc[2], fma(q[2], c.psc[1], q[1] * c.psc[0]));
        F[0] = coeff * q[0];
        F[1] = fma(a[1], c.psc[0], coeff * q[1]);
        F[2] = fma(a[1], c.psc[1], coeff * q[2]);
        F[3] = fma(a[1], c.psc[2], coeff * q[3]);
        F[4] = coeff * (q[4] + a[1]);
    }
    __device__ static inline void flux_rt(const double* q, int d, double* F) {
        double a[2];
        aux(q, a);
        const double coeff = a[0] * q[d + 1];
        F[0] = coeff * q[0];
        F[1] = coeff * q[1];
        F[2] = coeff * q[2];
        F[3] = coeff * q[3];
        F[4] = coeff * q[4] + coeff * a[1];
        F[d + 1] += a[1];
    }
    __device__ static inline double maxeig(const double* q, int d) {
        const double irho = 1.0 / fabs(q[0]);
        const double p = (GAMMA - 1) * (q[4] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
        const double c = sqrt(GAMMA * fabs(p) * irho);
#ifdef EXA_EULER_EIG_SELECT      // (r5 experiment: selects instead of the run-time index)
        const double un = (d == 0 ? q[1] : (d == 1 ? q[2] : q[3])) * irho;
#else
        const double un = q[d + 1] * irho;
#endif
        return fmax(fabs(un - c), fabs(un + c));
    }
    __device__ static inline double maxeig_fast(const double* q, int d) {
        const double irho = fast_rcp(fabs(q[0]));
        const double p = (GAMMA - 1) * (q[4] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
        const double c = fast_sqrt(GAMMA * fabs(p) * irho);
        const double un = (d == 0 ? q[1] : (d == 1 ? q[2] : q[3])) * irho;
        return fmax(fabs(un - c), fabs(un + c));
    }
    // Per-volume scalars for the FV patch kernels (corrected Rusanov): every volume's flux and eigenvalue are needed by
    // its own update and by its 2*dim neighbours', so 1/rho, p and the sound speed are computed ONCE per volume (fast
    // reciprocal / square root: <= 1e-15 relative, tolerance 1e-10) and kept beside the state.
    static constexpr int NFVAUX = 3;
    __device__ static inline void fv_aux(const double* q, double* a) {
        const double irho = fast_rcp(q[0]);
        const double p = (GAMMA - 1) * fma(-0.5 * irho, fma(q[3], q[3], fma(q[2], q[2], q[1] * q[1])), q[4]);
        a[0] = irho;
        a[1] = p;
        a[2] = fast_sqrt(GAMMA * fabs(p) * fabs(irho));
    }
    template <int D> __device__ static inline void flux_fv(const double* q, const double* a, double* F) {
        const double coeff = a[0] * q[D + 1];                 // explicit fma: the FV unit is compiled without contraction (faithful mode)
        F[0] = q[D + 1];
        F[1] = D == 0 ? fma(coeff, q[1], a[1]) : coeff * q[1];
        F[2] = D == 1 ? fma(coeff, q[2], a[1]) : coeff * q[2];
        F[3] = D == 2 ? fma(coeff, q[3], a[1]) : coeff * q[3];
        F[4] = coeff * (q[4] + a[1]);
    }
    template <int D> __device__ static inline double maxeig_fv(const double* q, const double* a) {
        return fabs(q[D + 1] * a[0]) + a[2];              // max(|u_n - c|, |u_n + c|) = |u_n| + c
    }
};

// Linear advection of NVARS variables with a fixed velocity (known-answer tests).
template <int NVARS> struct Advection {
    static constexpr int NV = NVARS;
    static constexpr int NFLUX = NVARS;
    static constexpr int NAUX = 1;     // unused slot (keeps array sizes non-zero)
    static constexpr int MAXDIM = 3;
    __device__ static inline double vel(int d) { return d == 0 ? 1.0 : (d == 1 ? 0.5 : -0.75); }
    __device__ static inline void aux(const double*, double* a) { a[0] = 0.0; }
    __device__ static inline void aux_fast(const double*, double* a) { a[0] = 0.0; }
    template <int D> __device__ static inline void flux(const double* q, const double*, double* F) {
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] = vel(D) * q[v];
    }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double*, double sc, double* F) {
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] = (vel(D) * sc) * q[v];
    }
    struct Dir { double c; };
    __device__ static inline void dir_init(Dir& c, int d, double sc) { c.c = vel(d) * sc; }
    __device__ static inline void flux_scaled_dir(const double* q, const double*, const Dir& c, double* F) {
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] = c.c * q[v];
    }
    __device__ static inline void flux_rt(const double* q, int d, double* F) {
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] = vel(d) * q[v];
    }
    __device__ static inline double maxeig(const double*, int d) { return fabs(vel(d)); }
    __device__ static inline double maxeig_fast(const double*, int d) { return fabs(vel(d)); }
};

}  // namespace exa
