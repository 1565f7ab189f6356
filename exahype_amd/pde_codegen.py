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
import re
import shutil
import subprocess

import sympy
from sympy.printing.c import C99CodePrinter

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
USER_DIR = os.path.join(HERE, "_user")


class _DevicePrinter(C99CodePrinter):
    """C for the device: small integer powers as products, 1/x as a division -- or, fast=True (the ADER-DG kernels, tolerance 1e-10), as
    exa::fast_rcp (v_rcp_f64 + one Newton step, <= 11 ulp) and square roots as exa::fast_sqrt (v_rsq_f64 + two Goldschmidt steps), the
    sequences the built-in term sets use (exa_pde.hpp)."""

    def __init__(self, fast=False):
        super().__init__()
        self.fast = fast

    def _rcp(self, x):
        return ("exa::fast_rcp(%s)" if self.fast else "(1.0/(%s))") % x

    def _print__Rcp(self, expr):
        return "exa::fast_rcp(%s)" % self._print(expr.args[0])

    def _print__Sqrt(self, expr):
        return "exa::fast_sqrt(%s)" % self._print(expr.args[0])

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer and 2 <= int(e) <= 4:
            return "(" + "*".join(["(%s)" % self._print(b)] * int(e)) + ")"
        if e == -1:
            return self._rcp(self._print(b))
        if e.is_Integer and -4 <= int(e) <= -2:
            return self._rcp("*".join(["(%s)" % self._print(b)] * (-int(e))))
        if self.fast and e == sympy.Rational(1, 2):
            return "exa::fast_sqrt(%s)" % self._print(b)
        if self.fast and e == sympy.Rational(-1, 2):
            return "exa::fast_rcp(exa::fast_sqrt(%s))" % self._print(b)
        return super()._print_Pow(expr)


class _Rcp(sympy.Function):
    """1/x by the fast sequence (printing only)"""
    nargs = 1


class _Sqrt(sympy.Function):
    nargs = 1


def _fast_forms(e):
    """Negative and half-integer powers as _Rcp / _Sqrt calls, so that the printer never emits an IEEE division (C99CodePrinter prints a
    product with negative powers as a quotient) and the common-subexpression pass shares the reciprocals."""
    def conv(p):
        b, ex = p.base, p.exp
        if ex.is_Integer and ex < 0:
            return _Rcp(b ** (-ex))
        if ex == sympy.Rational(1, 2):
            return _Sqrt(b)
        if ex == sympy.Rational(-1, 2):
            return _Rcp(_Sqrt(b))
        if ex.is_Rational and ex.q == 2:
            r = _Sqrt(b) * b ** ((abs(ex.p) - 1) // 2)
            return r if ex > 0 else _Rcp(r)
        return p
    return e.replace(lambda x: x.is_Pow and (x.exp.is_negative or (x.exp.is_Rational and x.exp.q == 2)), conv)


def _weighted_ops(e):
    """Vector-ALU instructions an expression costs, roughly: add / mul / fma-able pair 1, reciprocal or square root 8, other functions 20."""
    if e.is_Atom:
        return 0
    c = sum(_weighted_ops(a) for a in e.args)
    if e.is_Mul:
        return c + len(e.args) - 1 - (1 if e.args[0] == -1 else 0)                  # (a sign is an operand modifier)
    if e.is_Add:                                                   # a product term's last multiply fuses with an add (v_fma_f64)
        return c + len(e.args) - 1 - min(sum(1 for a in e.args if a.is_Mul and not (len(a.args) == 2 and a.args[0] == -1)), len(e.args) - 1)
    if e.is_Pow:
        ex = e.exp
        if ex.is_Integer:
            return c + (abs(int(ex)) - 1) + (8 if ex < 0 else 0)
        if ex in (sympy.Rational(1, 2), sympy.Rational(-1, 2)):
            return c + (8 if ex > 0 else 16)
        return c + 40
    if isinstance(e, (sympy.Abs, sympy.Max, sympy.Min)):
        return c + len(e.args) - 1
    if isinstance(e, (_Rcp, _Sqrt)):
        return c + 8
    return c + 20


def _arity(fn):
    import inspect
    return len([p for p in inspect.signature(fn).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)])


class SympyPDE:
    def __init__(self, n_vars, flux, max_eigenvalue, max_dim=3, name="user", source=None, ncp=None, max_aux=None):
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
        self.max_aux = max_aux                      # cached flux scalars per node: None = as many as pay and fit (see _analyse)
        self.dg_flags = None                        # development aid: compiler flags of the stage-A units instead of build.DG_SCHED
        self._aux = None
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
    def _block(self, exprs, targets, indent, fast=False, extra=None):
        pr = _DevicePrinter(fast)
        subs = {s: sympy.Symbol("q[%d]" % i) for i, s in enumerate(self.q)}
        subs.update({s: sympy.Symbol("x[%d]" % i) for i, s in enumerate(self.x)})
        subs.update({s: sympy.Symbol("dq[%d]" % i) for i, s in enumerate(self.dq)})
        subs.update(extra or {})
        exprs = [sympy.sympify(e).xreplace(subs) for e in exprs]
        if fast:
            exprs = [_fast_forms(e) for e in exprs]
        repl, red = sympy.cse(exprs, symbols=sympy.numbered_symbols("t_"))
        lines = ["%sconst double %s = %s;" % (indent, pr.doprint(a), pr.doprint(b)) for a, b in repl]
        lines += ["%s%s = %s;" % (indent, t, pr.doprint(e)) for t, e in zip(targets, red)]
        return "\n".join(lines)

    def _eig_block(self, exprs_by_dir, indent, fast=False):
        """Body of an eigenvalue member with a run-time, per-LANE normal `d` (the Riemann kernels pack tasks of all faces into a wave): ONE straight-line
        block -- what the directions share is computed once (common-subexpression pass over all of them) -- and, where the directions' expressions have the
        same shape and differ only in leaves (Euler: |q[1 + d] / rho| + c), the differing leaves are picked by selects in FRONT of the arithmetic, so the
        block costs what one direction costs; otherwise every direction is evaluated and the result picked.  (A `switch (d)` ran each direction's
        divisions and square root in turn under masks: r4, +12 % on stage B, an HBM-bound kernel.)"""
        md = len(exprs_by_dir)
        pr = _DevicePrinter(fast)
        subs = {s: sympy.Symbol("q[%d]" % i) for i, s in enumerate(self.q)}
        subs.update({s: sympy.Symbol("x[%d]" % i) for i, s in enumerate(self.x)})
        exprs = [sympy.sympify(e).xreplace(subs) for e in exprs_by_dir]
        if fast:
            exprs = [_fast_forms(e) for e in exprs]
        repl, red = sympy.cse(exprs, symbols=sympy.numbered_symbols("t_"))
        lines = ["%sconst double %s = %s;" % (indent, pr.doprint(a), pr.doprint(b)) for a, b in repl]
        pick = lambda names: "".join("d == %d ? %s : (" % (d, n) for d, n in enumerate(names[:-1])) + names[-1] + ")" * (len(names) - 1)
        sels = []

        def unify(es):
            if all(e == es[0] for e in es):
                return es[0]
            if all(e.is_Atom for e in es):
                sym = sympy.Symbol("s_%d" % len(sels))
                sels.append((sym, list(es)))
                return sym
            if any(e.is_Atom for e in es) or len({(e.func, len(e.args)) for e in es}) != 1:
                raise ValueError("shapes differ")
            return es[0].func(*[unify([e.args[k] for e in es]) for k in range(len(es[0].args))])
        try:
            tmpl = unify(list(red)) if md > 1 else red[0]
            ok = all(tmpl.xreplace({sym: leaves[d] for sym, leaves in sels}) == red[d] for d in range(md))      # (Add / Mul may have re-ordered: check)
        except ValueError:
            ok = False
        def leaf_select(leaves):
            # leaves that are entries of the state: a run-time INDEX (q[d + 1]) instead of a select of doubles -- in the Riemann kernel of 3-D p = 5 the
            # select form costs 6 - 8 VGPRs and with them the eighth resident wave per SIMD (16.5 against 15.7 ms per 128^3 launch, r5)
            m = [re.fullmatch(r"q\[(\d+)\]", str(l)) for l in leaves]
            if all(m) and len(leaves) > 1:
                idx = [int(x.group(1)) for x in m]
                step = idx[1] - idx[0]
                if all(idx[d] == idx[0] + d * step for d in range(len(idx))) and step != 0:
                    return "q[%d + %s]" % (idx[0], "d" if step == 1 else "%d * d" % step)
                return "q[%s]" % pick([str(i) for i in idx])
            return pick([pr.doprint(l) for l in leaves])
        if ok:
            lines += ["%sconst double %s = %s;" % (indent, sym, leaf_select(leaves)) for sym, leaves in sels]
            lines.append("%sconst double lam = %s;" % (indent, pr.doprint(tmpl)))
        else:
            lines += ["%sconst double lam%d = %s;" % (indent, d, pr.doprint(e)) for d, e in enumerate(red)]
            lines.append("%sconst double lam = %s;" % (indent, pick(["lam%d" % d for d in range(md)])))
        lines.append("%sreturn d < %d ? lam : 0.0;" % (indent, md))
        return "\n".join(lines)

    @staticmethod
    def _block_ops(exprs):
        repl, red = sympy.cse(list(exprs))
        return sum(_weighted_ops(b) for _, b in repl) + sum(_weighted_ops(e) for e in red)

    def _best_form(self, exprs, gens):
        """The cheapest (weighted instruction count after common-subexpression elimination) of a few algebraic forms of a block of
        expressions: as written, expanded, with common terms factored, and collected in the cached scalars / scale symbols `gens` with
        factored coefficients (that one finds `a0 (n0 q1 + n1 q2 + n2 q3) q1 + n0 a1` in the direction-masked Euler flux)."""
        exprs = [sympy.sympify(e) for e in exprs]
        alts = []
        for e in exprs:
            a = [e]
            for f in (sympy.expand, sympy.factor_terms,
                      lambda x: sympy.collect(sympy.expand(x), gens, func=sympy.factor) if gens else sympy.factor(x)):
                try:
                    g = f(e)
                    if g not in a:
                        a.append(g)
                except Exception:                                  # (a form SymPy cannot build for this expression is simply not a candidate)
                    pass
            alts.append(a)
        # whole block in one form, then expression by expression while the count falls (the forms interact through shared sub-expressions)
        nforms = max(len(a) for a in alts)
        best = min(([a[min(k, len(a) - 1)] for a in alts] for k in range(nforms)), key=self._block_ops)
        score = self._block_ops(best)
        for _ in range(2):
            for i, a in enumerate(alts):
                for g in a:
                    trial = best[:i] + [g] + best[i + 1:]
                    sc_ = self._block_ops(trial)
                    if sc_ < score:
                        best, score = trial, sc_
        return best

    def _cap_aux(self):
        """Cached scalars the tuned kernels have LDS for: exa_dg_reg.hpp (N = 6: two cells of (4 NV + NAUX) slots of 2 x 217 doubles + 2 KB in
        160 KB) and exa_dg_m8.hpp (N = 8: (3 NV + NAUX) slots of 2 x 560 doubles + 6 KB); a kernel that does not fit NV at all does not count."""
        caps = [c for c in ((163840 - 2048) // (2 * 8 * 434) - 4 * self.n_vars, (163840 - 6144) // (8 * 1120) - 3 * self.n_vars) if c >= 0]
        return min(caps + [4])

    def _analyse(self):
        """Pick the per-node scalars worth caching (struct member `aux`): sub-expressions of the fluxes that at least two directions share and
        that cost more to recompute than an LDS load (greedy by saved instructions, bounded by the kernels' LDS).  For Euler written the
        obvious way this finds 1/rho and p -- what the hand-written exa::Euler caches."""
        if getattr(self, "_aux", None) is not None:
            return
        n = self.n_vars
        flat = [e for d in range(self.max_dim) for e in self.flux_exprs[d]]
        repl, red = sympy.cse(flat, symbols=sympy.numbered_symbols("c_"))
        defs = dict(repl)
        order = {t: i for i, (t, _) in enumerate(repl)}

        def deps(e):
            return [s for s in e.free_symbols if s in defs]
        uses = {t: set() for t in defs}
        for i, r in enumerate(red):
            stack, seen = deps(r), set()
            while stack:
                t = stack.pop()
                if t in seen:
                    continue
                seen.add(t)
                uses[t].add(i // n)
                stack += deps(defs[t])

        def full(e):
            while True:
                fs = {s: defs[s] for s in e.free_symbols if s in defs}
                if not fs:
                    return e
                e = e.xreplace(fs)
        # shared temporaries that differ by a cheap term are chained (SymPy distributes a numeric factor over a sum, so `E + p` with
        # p = 0.4 (E - ...) arrives as `1.4 E - ...`, unrelated to p as far as the sub-expression pass can see): t_j := t_i + (t_j - t_i)
        shared = [t for t, _ in repl if len(uses[t]) >= 2]
        for jx, tj in enumerate(shared):
            for ti in shared[:jx]:
                if ti in defs[tj].free_symbols or _weighted_ops(defs[ti]) < 1:
                    continue
                diff = sympy.expand(full(defs[tj]) - full(defs[ti]))
                # (1.4 - 0.4 is 0.9999999999999999 in binary: coefficients within rounding of a NON-ZERO integer are that integer -- a relative test: a small
                # physical constant such as 1e-15 is a coefficient of the user's PDE, not a rounding residue, and must not be snapped to 0)
                diff = diff.xreplace({f: sympy.Integer(round(f)) for f in diff.atoms(sympy.Float)
                                      if round(f) != 0 and abs(f - round(f)) <= 4e-16 * abs(f)})
                if _weighted_ops(diff) <= 1 and not (diff.free_symbols & set(defs)):
                    defs[tj] = ti + diff
                    break

        def reach(exprs, stop, own=()):
            """weighted instructions of `exprs` plus every temporary they need (each once), not descending into `stop` (`own`: temporaries
            whose definitions are among `exprs`)"""
            seen, total, stack = set(own), 0, list(exprs)
            while stack:
                e = stack.pop()
                total += _weighted_ops(e)
                for s in deps(e):
                    if s not in stop and s not in seen:
                        seen.add(s)
                        stack.append(defs[s])
            return total

        def objective(chosen):
            """per node and level: the flux of every direction (each recomputes what is not cached) + the owner's aux block + the LDS
            round trip of the cached scalars (one store, one load per direction)"""
            per_dir = sum(reach(red[d * n:(d + 1) * n], chosen) for d in range(self.max_dim))
            return per_dir + reach([defs[t] for t in chosen], (), chosen) + (self.max_dim + 1) * len(chosen)
        chosen = []
        cap = self._cap_aux() if self.max_aux is None else self.max_aux
        while len(chosen) < cap:
            now = objective(chosen)
            best = min(((objective(chosen + [t]), order[t], t) for t in defs if t not in chosen and len(uses[t]) >= 2), default=None)
            if best is None or best[0] >= now:
                break
            chosen.append(best[2])
        chosen.sort(key=lambda t: order[t])
        self._aux_syms = [sympy.Symbol("a[%d]" % k) for k in range(len(chosen))]
        to_a = dict(zip(chosen, self._aux_syms))

        def inline(e, keep):
            while True:
                fs = {s: defs[s] for s in e.free_symbols if s in defs and s not in keep}
                if not fs:
                    return e
                e = e.xreplace(fs)
        self._aux = [inline(defs[t], ()) for t in chosen]                                    # in the state alone
        self._flux_a = [[inline(red[d * n + v], chosen).xreplace(to_a) for v in range(n)] for d in range(self.max_dim)]   # in the state + cached scalars

    def n_aux(self):
        self._analyse()
        return len(self._aux)

    def source(self):
        n = self.n_vars
        flux_cases, eig_cases = [], []
        for d in range(self.max_dim):
            flux_cases.append("        case %d: {\n%s\n        } break;" % (d, self._block(self.flux_exprs[d], ["F[%d]" % v for v in range(n)], "            ")))
        # the eigenvalue members take a run-time, per-LANE normal in the Riemann kernels: ONE straight-line block for all directions (what they share is
        # computed once), the lane's direction picked by selects -- see _source_tuned (a `switch (d)` ran every direction's divisions and square root in turn)
        md = self.max_dim
        eig_cases = [self._eig_block([self.eig_exprs[d] for d in range(md)], "        ")]
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
            for fast in (False, True):
                cases = ["        case %d: {\n%s\n        } break;" % (d, self._block(self.ncp_exprs[d], ["out[%d]" % v for v in range(n)], "            ", fast=fast))
                         for d in range(self.max_dim)]
                nm = "ncp_xt" if self.uses_xt else "ncp"
                sig = ("%s%s(const double* q, const double* dq, const double* x, double t, int d, double* out)" if self.uses_xt
                       else "%s%s(const double* q, const double* dq, int d, double* out)") % (nm, "_fast" if fast else "")
                src_member += ("%s    __device__ static inline void %s {\n        switch (d) {\n%s\n"
                               "        default:\n            for (int v = 0; v < NV; v++) out[v] = 0.0;\n        }\n    }\n"
                               % ("" if fast else "    static constexpr bool HAS_NCP = true;\n", sig, "\n".join(cases)))
        if self.uses_xt:
            fast_cases = ["        case %d: {\n%s\n        } break;" % (d, self._block(self.flux_exprs[d], ["F[%d]" % v for v in range(n)], "            ", fast=True))
                          for d in range(self.max_dim)]
            src_member += ("    // (the ADER-DG kernels' twin of flux_xt: reciprocals / square roots by the fast sequences, tolerance 1e-10)\n"
                           "    __device__ static inline void flux_xt_fast(const double* q, const double* x, double t, int d, double* F) {\n        switch (d) {\n%s\n"
                           "        default:\n            for (int v = 0; v < NV; v++) F[v] = 0.0;\n        }\n    }\n" % "\n".join(fast_cases))
            return self._source_xt(flux_cases, eig_cases, src_member)
        return self._source_tuned(flux_cases, eig_cases, src_member)

    def _tuned_flux_members(self):
        """aux / aux_fast bodies, flux<D> / flux_scaled<D> branches and the Dir member for a flux that depends on the state alone"""
        self._analyse()
        n, na, md = self.n_vars, len(self._aux), self.max_dim
        sc = sympy.Symbol("sc")
        A = self._aux_syms
        ind = "            "
        Fv = ["F[%d]" % v for v in range(n)]
        aux_t = ["a[%d]" % k for k in range(na)]
        aux_ieee = self._block(self._aux, aux_t, "        ") if na else ""
        aux_fast = self._block(self._aux, aux_t, "        ", fast=True) if na else ""
        flux_d, flux_sc = [], []
        for d in range(md):
            flux_d.append("        if constexpr (D == %d) {\n%s\n        }" % (d, self._block(self._best_form(self._flux_a[d], A), Fv, ind, fast=True)))
            scaled = self._best_form([sc * e for e in self._flux_a[d]], A + [sc])
            flux_sc.append("        if constexpr (D == %d) {\n%s\n        }" % (d, self._block(scaled, Fv, ind, fast=True)))
        dir_member = ""
        self.dir_form = None
        if md == 3:
            nn = list(sympy.symbols("n0:3"))
            masked = self._best_form([sum(nn[d] * self._flux_a[d][v] for d in range(3)) for v in range(n)], A + nn)
            ops_m = self._block_ops(masked)
            ops_d = max(self._block_ops([sc * e for e in self._flux_a[d]]) for d in range(3))
            self.dir_form = {"masked_ops": ops_m, "one_direction_ops": ops_d}
            # two of the four derive waves of exa_dg_reg.hpp hold pencils of two directions: the switch costs them 2 x the flux
            if ops_m <= 1.8 * ops_d:
                body = self._block(masked, Fv, "        ", fast=True, extra={nn[d]: sympy.Symbol("c.n[%d]" % d) for d in range(3)})
                dir_member = ("    // normal chosen per LANE: n[k] = scale for the lane's own direction, 0 for the others (weighted %d instructions against %d for one\n"
                              "    // compile-time direction)\n"
                              "    struct Dir { double n[3]; };\n"
                              "    __device__ static inline void dir_init(Dir& c, int d, double sc) {\n"
                              "        c.n[0] = d == 0 ? sc : 0.0;\n        c.n[1] = d == 1 ? sc : 0.0;\n        c.n[2] = d == 2 ? sc : 0.0;\n    }\n"
                              "    __device__ static inline void flux_scaled_dir(const double* q, const double* a, const Dir& c, double* F) {\n%s\n    }\n"
                              % (ops_m, ops_d, body))
        return {"na": na, "aux_ieee": aux_ieee, "aux_fast": aux_fast, "flux_d": flux_d, "flux_sc": flux_sc, "dir_member": dir_member}

    def _source_tuned(self, flux_cases, eig_cases, src_member):
        """Term set of the state alone (no position / time, no ncp): the interface the tuned ADER-DG kernels are written against, with what
        the hand-written exa::Euler has -- per-node cached scalars shared by the directions (aux / aux_fast), reciprocals and square roots by
        the fast sequences in the members only the ADER-DG kernels call, the scale folded into the flux (flux_scaled), and the flux for a
        per-LANE normal as straight-line code over per-lane masks (Dir / dir_init / flux_scaled_dir; exa_dg_reg.hpp) where that form costs
        less than the divergent switch it replaces."""
        m = self._tuned_flux_members()
        n, na, md = self.n_vars, m["na"], self.max_dim
        ind = "            "
        aux_ieee, aux_fast, flux_d, flux_sc, dir_member = m["aux_ieee"], m["aux_fast"], m["flux_d"], m["flux_sc"], m["dir_member"]
        # maxeig_fast: stage B calls it with a per-LANE normal (tasks (cell, face, node) packed densely over a wave), so a `switch (d)` runs every
        # direction's block in turn under masks -- with its own reciprocal and square root each (r4: 17.5 against 15.5 ms per 128^3 launch of stage B, an
        # HBM-bound kernel).  Straight-line code instead: the directions' expressions in ONE block (what they share -- 1/rho, the pressure, the sound
        # speed -- is computed once by the common-subexpression pass), the lane's direction picked by selects; where d is a compile-time constant after
        # inlining, the unused directions fold away.
        eig_fast = self._eig_block([self.eig_exprs[d] for d in range(md)], "        ", fast=True)
        return """// generated by exahype_amd/pde_codegen.py from SymPy expressions -- user PDE term set "%s"
#pragma once
#include <hip/hip_runtime.h>
#include "exa_pde.hpp"
namespace exa {
struct UserPDE {
    static constexpr int NV = %d;
    static constexpr int NFLUX = %d;
    static constexpr int NAUX = %d;             // per-node scalars shared by the directions, found in the expressions (pde_codegen._analyse)
    static constexpr int MAXDIM = %d;
    __device__ static inline void aux(const double* q, double* a) {
%s
    }
    __device__ static inline void aux_fast(const double* q, double* a) {
%s
    }
    __device__ static inline void flux_rt(const double* q, int d, double* F) {
        switch (d) {
%s
        default:
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
    template <int D> __device__ static inline void flux(const double* q, const double* a, double* F) {
%s
        if constexpr (D >= MAXDIM) {
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double* a, double sc, double* F) {
%s
        if constexpr (D >= MAXDIM) {
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
%s    __device__ static inline double maxeig(const double* q, int d) {
%s
    }
    __device__ static inline double maxeig_fast(const double* q, int d) {
%s
    }
%s};
}  // namespace exa
""" % (self.name, n, n, na, md, aux_ieee, aux_fast, "\n".join(flux_cases), "\n".join(flux_d), "\n".join(flux_sc), dir_member,
       "\n".join(eig_cases), eig_fast, src_member)

    def _source_xt(self, flux_cases, eig_cases, src_member):
        """Term set whose expressions contain the volume centre x or the time t: the *_xt members carry them (fv_rusanov.hip uses them
        where HAS_XT is set).  Where the FLUX itself does not depend on x, t (a position- / time-dependent source or eigenvalue beside an ordinary
        flux -- gravity, forcing) the ADER-DG kernels get the tuned flux interface of _source_tuned (cached scalars, scaled and per-lane-normal
        forms; FLUX_XT = false) and hand the coordinates to the source only; otherwise the plain members evaluate flux_xt at x = 0, t = 0 and
        exist only so that the common interface is complete."""
        n = self.n_vars
        xt = set(self.x) | {self.t}
        flux_xt_dep = any(e.free_symbols & xt for f in self.flux_exprs for e in f)
        if flux_xt_dep:
            naux, members = 0, """    static constexpr bool FLUX_XT = true;
    __device__ static inline void aux(const double*, double*) {}
    __device__ static inline void aux_fast(const double*, double*) {}
    template <int D> __device__ static inline void flux(const double* q, const double*, double* F) { flux_rt(q, D, F); }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double*, double sc, double* F) {
        flux_rt(q, D, F);
#pragma unroll
        for (int v = 0; v < NV; v++) F[v] *= sc;
    }
"""
        else:
            m = self._tuned_flux_members()
            naux = m["na"]
            members = """    static constexpr bool FLUX_XT = false;        // the flux sees the state alone: tuned forms below, coordinates reach the source / eigenvalue only
    __device__ static inline void aux(const double* q, double* a) {
%s
    }
    __device__ static inline void aux_fast(const double* q, double* a) {
%s
    }
    template <int D> __device__ static inline void flux(const double* q, const double* a, double* F) {
%s
        if constexpr (D >= MAXDIM) {
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
    template <int D> __device__ static inline void flux_scaled(const double* q, const double* a, double sc, double* F) {
%s
        if constexpr (D >= MAXDIM) {
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
%s""" % (m["aux_ieee"], m["aux_fast"], "\n".join(m["flux_d"]), "\n".join(m["flux_sc"]), m["dir_member"])
        return """// generated by exahype_amd/pde_codegen.py from SymPy expressions -- user PDE term set "%s" (terms depend on position / time)
#pragma once
#include <hip/hip_runtime.h>
#include "exa_pde.hpp"
namespace exa {
struct UserPDE {
    static constexpr int NV = %d;
    static constexpr int NFLUX = %d;
    static constexpr int NAUX = %d;
    static constexpr int MAXDIM = %d;
    static constexpr bool HAS_XT = true;
    __device__ static inline void flux_xt(const double* q, const double* x, double t, int d, double* F) {
        switch (d) {
%s
        default:
            for (int v = 0; v < NV; v++) F[v] = 0.0;
        }
    }
    __device__ static inline double maxeig_xt(const double* q, const double* x, double t, int d) {
%s
    }
    __device__ static inline void flux_rt(const double* q, int d, double* F) { const double x0[3] = {0.0, 0.0, 0.0}; flux_xt(q, x0, 0.0, d, F); }
%s    __device__ static inline double maxeig(const double* q, int d) { const double x0[3] = {0.0, 0.0, 0.0}; return maxeig_xt(q, x0, 0.0, d); }
    __device__ static inline double maxeig_fast(const double* q, int d) { return maxeig(q, d); }
%s};
}  // namespace exa
""" % (self.name, n, n, naux, self.max_dim, "\n".join(flux_cases), "\n".join(eig_cases), members, src_member)

    def key(self):
        h = hashlib.sha256(self.source().encode())
        for f in ("dg_inst.hip", "fv_rusanov.hip", "exa_dg_kernels.hpp", "exa_dg_stream.hpp", "exa_dg_reg.hpp", "exa_dg_fused.hpp",
                  "exa_dg_common.hpp", "exa_launch.hpp", "exa_pde.hpp", "exa_dg_plain.hpp", "exa_dg_m8.hpp"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
        h.update(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build.py"), "rb").read())      # (compiler flags)
        h.update(repr(self.dg_flags).encode())
        h.update(os.environ.get("EXA_EXTRA_FLAGS", "").encode())       # (development builds with extra -D macros: a library of their own)
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
                  "-Wno-pass-failed", "-I", CSRC, "-DEXA_PDE_ID=100", '-DEXA_USER_PDE_HEADER="%s"' % hdr] + os.environ.get("EXA_EXTRA_FLAGS", "").split()
        sched = self.dg_flags if self.dg_flags is not None else _build.DG_SCHED
        units = [("fv_rusanov.hip", "fv.o", ["-ffp-contract=off"]), ("dg_inst.hip", "dg2.o", ["-DEXA_DIM=2", "-DEXA_UNIT_A"] + sched),
                 ("dg_inst.hip", "dg2b.o", ["-DEXA_DIM=2", "-DEXA_UNIT_B"])]
        if self.max_dim >= 3:
            units += [("dg_inst.hip", "dg3.o", ["-DEXA_DIM=3", "-DEXA_UNIT_A"] + sched), ("dg_inst.hip", "dg3b.o", ["-DEXA_DIM=3", "-DEXA_UNIT_B"])]
        procs = [(o, subprocess.Popen(common + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(d, o)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_build.compiler_env())) for src, o, extra in units]
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
