"""User PDE terms from SymPy expressions, compiled for the device.

In the reference the PDE terms of a kernel are opaque named functions (`TypedFunction`), resolved at
link time to hand-written host C++ (`Unit test/Functions.cpp:9-66`) or Peano solver methods
(`examples/kernel-generator.py:22-23`).  A HIP kernel cannot call host code, so a user system that is
not one of the built-in term sets is given as SymPy expressions instead:

    q = SympyPDE.state(3)                                     # symbols q0, q1, q2
    swe = SympyPDE(3, flux=lambda q, d: [...], max_eigenvalue=lambda q, d: ...)
    HIPPrinter(kernel, pde=swe)                               # or AderDgSolver(..., pde=swe.register())

`source()` is a `struct exa::UserPDE` with the interface of exahype_amd/csrc/exa_pde.hpp (common
sub-expressions eliminated per direction); `build()` compiles the kernel units for it with hipcc into
a side library (cached by content hash under exahype_amd/_user/); `register()` hands it to
libexahype_hip.so (`exa_register_pde`) and returns the pde id.
"""
import hashlib
import os
import shutil
import subprocess

import sympy
from sympy.printing.c import C99CodePrinter

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
USER_DIR = os.path.join(HERE, "_user")


class _DevicePrinter(C99CodePrinter):
    """C for the device: small integer powers as products, 1/x as a division."""

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer and 2 <= int(e) <= 4:
            return "(" + "*".join(["(%s)" % self._print(b)] * int(e)) + ")"
        if e == -1:
            return "(1.0/(%s))" % self._print(b)
        if e.is_Integer and -4 <= int(e) <= -2:
            return "(1.0/(" + "*".join(["(%s)" % self._print(b)] * (-int(e))) + "))"
        return super()._print_Pow(expr)


def _arity(fn):
    import inspect
    return len([p for p in inspect.signature(fn).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)])


class SympyPDE:
    def __init__(self, n_vars, flux, max_eigenvalue, max_dim=3, name="user", source=None, ncp=None):
        """flux(q, d) -> n_vars expressions, max_eigenvalue(q, d) -> one, in the state symbols q; d = 0-based normal.
        source(q) -> n_vars expressions (optional): the algebraic source S(q) of q_t + div F(q) = S(q) -- the hook the
        reference's harness declares beside flux and maxEigenvalue (`Unit test/correctness_test.cpp:16-23`).  It enters the
        ADER-DG predictor, the time-averaged volume term and the corrected FV update; the faithful FV mode (the reference's statement list) has no source by construction.

        ncp(q, dq, d) (or ncp(q, dq, x, t, d)) -> n_vars expressions (optional): the non-conservative product B_d(q) dq of
        q_t + div F(q) + B(q) . grad q = S(q), dq = the jump (or gradient) of q along direction d -- the `ncp` slot of the kernel the
        harness targets (`Unit test/correctness_test.cpp:145-155`).  Built into the corrected FV Rusanov update (path-conservative jump
        term, half to either side of a face) and into ADER-DG (exa_dg_plain.hpp: B . grad q point-wise in the predictor and in the
        time-averaged update, the same jump term in the Riemann solve).

        Position and time: the harness declares the terms as flux / maxEigenvalue / sourceTerm(Q, x, h, t, dt, ...)
        (`Unit test/correctness_test.cpp:16-41`).  Callables that take them -- flux(q, x, t, d), max_eigenvalue(q, x, t, d),
        source(q, x, t), x = (x0, x1, x2) -- generate a term set with HAS_XT; the FV patch kernels hand it the volume centres
        (`exa_fv_time_step_device_oop`: cell centres and t; the in-place call: patches centred at the origin, t = 0), ADER-DG the node
        coordinates and level times (`exa_dg_plan_set_origin_time`).  Term sets with HAS_XT or an ncp run ADER-DG through a plain,
        untuned stage-A kernel (2-D N <= 8, 3-D N <= 6); the tuned kernels serve the others."""
        if not 1 <= n_vars <= 8:
            raise ValueError("n_vars must be 1..8")
        self.n_vars, self.max_dim, self.name = n_vars, max_dim, name
        self.q = self.state(n_vars)
        self.x = list(sympy.symbols("x0:3", real=True))
        self.t = sympy.Symbol("t", real=True)
        self.flux_exprs = []
        self.eig_exprs = []
        self.source_exprs = None
        fx = (lambda q, d: flux(q, self.x, self.t, d)) if _arity(flux) == 4 else flux
        ex = (lambda q, d: max_eigenvalue(q, self.x, self.t, d)) if _arity(max_eigenvalue) == 4 else max_eigenvalue
        if source is not None:
            self.source_exprs = [sympy.sympify(e) for e in (source(self.q, self.x, self.t) if _arity(source) == 3 else source(self.q))]
            if len(self.source_exprs) != n_vars:
                raise ValueError("source(q) must return %d expressions" % n_vars)
        for d in range(max_dim):
            f = [sympy.sympify(e) for e in fx(self.q, d)]
            if len(f) != n_vars:
                raise ValueError("flux(q, %d) must return %d expressions" % (d, n_vars))
            self.flux_exprs.append(f)
            self.eig_exprs.append(sympy.sympify(ex(self.q, d)))
        self.dq = list(sympy.symbols("dq0:%d" % n_vars, real=True))
        self.ncp_exprs = None
        if ncp is not None:
            nx = (lambda q, dq, d: ncp(q, dq, self.x, self.t, d)) if _arity(ncp) == 5 else ncp
            self.ncp_exprs = []
            for d in range(max_dim):
                e = [sympy.sympify(v) for v in nx(self.q, self.dq, d)]
                if len(e) != n_vars:
                    raise ValueError("ncp(q, dq, %d) must return %d expressions" % (d, n_vars))
                self.ncp_exprs.append(e)
        xt = set(self.x) | {self.t}
        every = [e for f in self.flux_exprs for e in f] + self.eig_exprs + (self.source_exprs or []) + [e for f in (self.ncp_exprs or []) for e in f]
        self.uses_xt = any(e.free_symbols & xt for e in every)
        self._lib = None
        self._id = None

    @staticmethod
    def state(n_vars):
        return list(sympy.symbols("q0:%d" % n_vars, real=True))

    # -- code generation -------------------------------------------------------------------------
    def _block(self, exprs, targets, indent):
        pr = _DevicePrinter()
        subs = {s: sympy.Symbol("q[%d]" % i) for i, s in enumerate(self.q)}
        subs.update({s: sympy.Symbol("x[%d]" % i) for i, s in enumerate(self.x)})
        subs.update({s: sympy.Symbol("dq[%d]" % i) for i, s in enumerate(self.dq)})
        repl, red = sympy.cse([e.subs(subs) for e in exprs], symbols=sympy.numbered_symbols("t_"))
        lines = ["%sconst double %s = %s;" % (indent, pr.doprint(a), pr.doprint(b)) for a, b in repl]
        lines += ["%s%s = %s;" % (indent, t, pr.doprint(e)) for t, e in zip(targets, red)]
        return "\n".join(lines)

    def source(self):
        n = self.n_vars
        flux_cases, eig_cases = [], []
        for d in range(self.max_dim):
            flux_cases.append("        case %d: {\n%s\n        } break;" % (d, self._block(self.flux_exprs[d], ["F[%d]" % v for v in range(n)], "            ")))
            eig_cases.append("        case %d: {\n%s\n            return lam;\n        }" % (d, self._block([self.eig_exprs[d]], ["const double lam"], "            ")))
        src_member = ""
        if self.source_exprs is not None:
            src_member = ("    static constexpr bool HAS_SOURCE = true;\n"
                          "    __device__ static inline void source%s {\n%s\n    }\n"
                          % ("_xt(const double* q, const double* x, double t, double* S)" if self.uses_xt else "(const double* q, double* S)",
                             self._block(self.source_exprs, ["S[%d]" % v for v in range(n)], "        ")))
            if self.uses_xt:
                src_member += ("    __device__ static inline void source(const double* q, double* S) { const double x0[3] = {0.0, 0.0, 0.0}; "
                               "source_xt(q, x0, 0.0, S); }\n")
        if self.ncp_exprs is not None:
            cases = ["        case %d: {\n%s\n        } break;" % (d, self._block(self.ncp_exprs[d], ["out[%d]" % v for v in range(n)], "            "))
                     for d in range(self.max_dim)]
            sig = "ncp_xt(const double* q, const double* dq, const double* x, double t, int d, double* out)" if self.uses_xt \
                else "ncp(const double* q, const double* dq, int d, double* out)"
            src_member += ("    static constexpr bool HAS_NCP = true;\n    __device__ static inline void %s {\n        switch (d) {\n%s\n"
                           "        default:\n            for (int v = 0; v < NV; v++) out[v] = 0.0;\n        }\n    }\n" % (sig, "\n".join(cases)))
        if self.uses_xt:
            return self._source_xt(flux_cases, eig_cases, src_member)
        return """// generated by exahype_amd/pde_codegen.py from SymPy expressions -- user PDE term set "%s"
#pragma once
#include <hip/hip_runtime.h>
namespace exa {
struct UserPDE {
    static constexpr int NV = %d;
    static constexpr int NFLUX = %d;
    static constexpr int NAUX = 1;
    static constexpr int MAXDIM = %d;
    __device__ static inline void aux(const double*, double* a) { a[0] = 0.0; }
    __device__ static inline void aux_fast(const double*, double* a) { a[0] = 0.0; }
    __device__ static inline void flux_rt(const double* q, int d, double* F) {
        switch (d) {
%s
        default:
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
    template <int D> __device__ static inline void flux(const double* q, const double*, double* F) { flux_rt(q, D, F); }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double*, double sc, double* F) {
        flux_rt(q, D, F);
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] *= sc;
    }
    __device__ static inline double maxeig(const double* q, int d) {
        switch (d) {
%s
        }
        return 0.0;
    }
    __device__ static inline double maxeig_fast(const double* q, int d) { return maxeig(q, d); }
%s};
}  // namespace exa
""" % (self.name, n, n, self.max_dim, "\n".join(flux_cases), "\n".join(eig_cases), src_member)

    def _source_xt(self, flux_cases, eig_cases, src_member):
        """Term set whose expressions contain the volume centre x or the time t: the *_xt members carry them (fv_rusanov.hip uses them
        where HAS_XT is set); the plain members evaluate at x = 0, t = 0 and exist only so that the common interface is complete."""
        n = self.n_vars
        return """// generated by exahype_amd/pde_codegen.py from SymPy expressions -- user PDE term set "%s" (terms depend on position / time)
#pragma once
#include <hip/hip_runtime.h>
namespace exa {
struct UserPDE {
    static constexpr int NV = %d;
    static constexpr int NFLUX = %d;
    static constexpr int NAUX = 1;
    static constexpr int MAXDIM = %d;
    static constexpr bool HAS_XT = true;
    __device__ static inline void aux(const double*, double* a) { a[0] = 0.0; }
    __device__ static inline void aux_fast(const double*, double* a) { a[0] = 0.0; }
    __device__ static inline void flux_xt(const double* q, const double* x, double t, int d, double* F) {
        switch (d) {
%s
        default:
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
    __device__ static inline double maxeig_xt(const double* q, const double* x, double t, int d) {
        switch (d) {
%s
        }
        return 0.0;
    }
    __device__ static inline void flux_rt(const double* q, int d, double* F) { const double x0[3] = {0.0, 0.0, 0.0}; flux_xt(q, x0, 0.0, d, F); }
    template <int D> __device__ static inline void flux(const double* q, const double*, double* F) { flux_rt(q, D, F); }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double*, double sc, double* F) {
        flux_rt(q, D, F);
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] *= sc;
    }
    __device__ static inline double maxeig(const double* q, int d) { const double x0[3] = {0.0, 0.0, 0.0}; return maxeig_xt(q, x0, 0.0, d); }
    __device__ static inline double maxeig_fast(const double* q, int d) { return maxeig(q, d); }
%s};
}  // namespace exa
""" % (self.name, n, n, self.max_dim, "\n".join(flux_cases), "\n".join(eig_cases), src_member)

    def key(self):
        h = hashlib.sha256(self.source().encode())
        for f in ("dg_inst.hip", "fv_rusanov.hip", "exa_dg_kernels.hpp", "exa_dg_stream.hpp", "exa_dg_reg.hpp", "exa_dg_fused.hpp",
                  "exa_dg_common.hpp", "exa_launch.hpp", "exa_pde.hpp", "exa_dg_plain.hpp", "exa_dg_m8.hpp"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
        h.update(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build.py"), "rb").read())      # (compiler flags)
        return h.hexdigest()[:16]

    # -- build + registration ----------------------------------------------------------------------
    def build(self, force=False):
        """Compile the kernel units for this term set (hipcc, gfx950) into a side library; returns its path."""
        from . import build as _build
        main_lib = _build.build()
        d = os.path.join(USER_DIR, self.key())
        so = os.path.join(d, "libexahype_user.so")
        if os.path.exists(so) and not force:
            return so
        with _build.build_lock(d):
            if os.path.exists(so) and not force:          # built by another process meanwhile (the link is renamed into place: complete)
                return so
            return self._build_locked(d, so, main_lib, _build)

    def _build_locked(self, d, so, main_lib, _build):
        hdr = os.path.join(d, "user_pde.hpp")
        with open(hdr, "w") as f:
            f.write(self.source())
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        common = [hipcc, "-O3", "-fPIC", "-std=c++17", "--offload-arch=%s" % _build.ARCH, "-Wno-unused-function",
                  "-Wno-pass-failed", "-I", CSRC, "-DEXA_PDE_ID=100", '-DEXA_USER_PDE_HEADER="%s"' % hdr]
        units = [("fv_rusanov.hip", "fv.o", ["-ffp-contract=off"]), ("dg_inst.hip", "dg2.o", ["-DEXA_DIM=2", "-DEXA_UNIT_A"] + _build.DG_SCHED),
                 ("dg_inst.hip", "dg2b.o", ["-DEXA_DIM=2", "-DEXA_UNIT_B"])]
        if self.max_dim >= 3:
            units += [("dg_inst.hip", "dg3.o", ["-DEXA_DIM=3", "-DEXA_UNIT_A"] + _build.DG_SCHED), ("dg_inst.hip", "dg3b.o", ["-DEXA_DIM=3", "-DEXA_UNIT_B"])]
        procs = [(o, subprocess.Popen(common + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(d, o)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for src, o, extra in units]
        for o, p in procs:
            _, err = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed for the user PDE (%s):\n%s" % (o, err[-3000:]))
        libdir = os.path.dirname(main_lib)
        _build.link_atomically([hipcc, "-shared", "-fPIC", "--offload-arch=%s" % _build.ARCH] + [os.path.join(d, o) for _, o, _ in units] +
                               ["-L", libdir, "-lexahype_hip", "-Wl,-rpath," + libdir], so)
        return so

    def register(self):
        """pde id (>= 100) of this term set in libexahype_hip.so (built and registered on first use)."""
        if self._id is None:
            from . import _lib
            self._id = _lib.register_pde(self.build())
        return self._id
