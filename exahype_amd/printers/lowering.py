"""Statement-by-statement lowering of a KernelBuilder state to HIP kernels -- the generic path of the printer seam.

The reference's `CPPPrinter` (`exahype/printers/CPPPrinter.py:84-137`) turns EVERY builder state into C++ text: temporaries on
the heap, then one loop nest per statement (patch outermost; the axes over the interior `[H, P+H)`, or over the full `[0, P+2H)`
along a statement's own direction when it has no `+-` offset; the variable loop `[0,1)`, `[0,n_real)` or `[0,n_real+n_aux)` by the
smallest struct a statement touches), AoS strides.  `HIPPrinter` dispatches the statement lists it RECOGNISES to fused hand-written
kernels; this module serves the others: one `__global__` kernel per statement with exactly the reference's ranges, strides and
expression text (taken from this package's text-compatible `CPPPrinter`, which is pinned byte for byte against the reference's
output), launched in statement order on one stream.  Temporaries live in HBM between the launches, as they live on the heap in the
reference -- ten sweeps instead of one, correct rather than fast; the recognised schemes keep their fused kernels.

A GPU runs a statement's loop nest in parallel and cannot be allowed to fault, so three things the reference's text leaves to
luck are checked here, and a state that fails one is refused (`UnrecognisedKernel`), never "fixed":
  * every array access of every statement lies inside its array for the whole loop range (the reference's own examples do
    not: `items[1]` is addressed with `patch - 1`);
  * a statement that writes an array reads that array only at the index it writes (no loop-carried dependence);
  * no OPAQUE function: the reference resolves `Flux(...)` to the user's host C++ at link time (`Unit test/Functions.h:2-4`), a device
    kernel cannot.  A function declared with a SymPy body (`kernel.function(name, ..., body=...)`) is lowered: the body becomes a
    `__device__` function with the reference's calling convention -- array arguments by address (`&Q[...]`: the callee sees the
    volume's variables from there on), a trailing array argument of a bare call is the out-parameter (`Flux(Q, normal, F)`,
    `exahype/printers/CPPPrinter.py:152-155`), a directional constant arrives as the integer normal -- and the statement keeps the
    reference's text (`CPPPrinter.py:204-276`).  `max(a, b)` of two scalars by address is `Unit test/Functions.cpp:64-66`.
Integer powers, which SymPy builds from `a*a` and the reference prints as the Python `a**2`, are printed as products.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import re
import shutil
import subprocess

import sympy
from sympy import Indexed, Integer, Symbol
from sympy.printing.str import StrPrinter
from sympy.core.function import AppliedUndef

from .CodePrinter import CodePrinter
from .CPPPrinter import CPPPrinter

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN_DIR = os.path.join(HERE, "_user")


class LoweringRefused(NotImplementedError):
    """The builder state cannot be lowered statement by statement (reason in the message)."""


def _default_max(a, b):
    """`max(double* a, double* b)` of the reference's user functions (`Unit test/Functions.cpp:64-66`)"""
    return sympy.Max(a[0], b[0])


class _CText(StrPrinter):
    """str(expr) as the reference prints it, except a**n (n = 2..4), printed as the product it came from."""

    def _print_Pow(self, expr, rational=False):
        b, e = expr.args
        if isinstance(e, Integer) and 2 <= int(e) <= 4:
            t = self._print(b)
            if not isinstance(b, (Indexed, Symbol)):
                t = "(%s)" % t
            return "*".join([t] * int(e))
        return super()._print_Pow(expr, rational=rational)

    def _print(self, expr, **kwargs):
        if isinstance(expr, Indexed):                     # Indexed carries its own _sympystr: keep the reference's form
            return str(expr)
        return super()._print(expr, **kwargs)


def _ctext(expr):
    return _CText().doprint(expr) if isinstance(expr, sympy.Basic) else str(expr)


class _CallText(_CText):
    """C text of a statement side that calls functions, printed from the tree: array elements with the reference's flat index
    (`CPPPrinter.Cppify` of the element alone), array arguments of a call by address -- once.  (The reference's text pass puts its `&` by
    substring replacement over the item names, `exahype/printers/CPPPrinter.py:230-236`: `Flux(Q_copy` becomes `Flux(&&Q_copy` as soon as
    `Q` is an item too, and an item named `F` gives `&Flux(`; what it MEANS is in its older output `Unit test/test.cpp:25`.)"""

    def __init__(self, cpp, functions):
        super().__init__()
        self._cpp, self._functions = cpp, functions

    def _print(self, expr, **kwargs):
        if isinstance(expr, Indexed):
            return self._cpp.Cppify(str(expr))
        if isinstance(expr, sympy.Function) and type(expr).__name__ in self._functions:
            return "%s(%s)" % (type(expr).__name__, ", ".join(("&" if isinstance(a, Indexed) else "") + self._print(a) for a in expr.args))
        return StrPrinter._print(self, expr, **kwargs)


class StatementLowering:
    """Analysis + HIP source + JIT build + launch of one builder state."""

    def __init__(self, kernel, function_name="time_step"):
        k = self.k = kernel
        self.name = function_name
        # the reference-compatible printer, for its range rule, its index flattening and its expression text only: its
        # constructor renders the whole function (and, like the reference, needs an input constant for the signature)
        self.cpp = CPPPrinter.__new__(CPPPrinter)
        CodePrinter.__init__(self.cpp, kernel, function_name)
        if not k.items:
            raise LoweringRefused("no item declared: nothing to pass to the kernel")
        if k.parents:
            raise LoweringRefused("items / constants with a parent object (%s) live inside a host C++ object the device cannot see"
                                  % ", ".join(sorted(k.parents)))
        full = k.patch_size + 2 * k.halo_size
        # arrays: name -> (doubles, leap, extent per axis), sizes as the reference allocates them (`CPPPrinter.py:62-76`)
        self.arrays = {}
        for name, it in k.all_items.items():
            if not isinstance(it, sympy.tensor.indexed.IndexedBase):
                continue
            leap = {0: 1, 1: k.n_real, 2: k.n_real + k.n_aux}[k.item_struct[name]]
            size = k.patch_size if (len(k.items) > 1 and name == k.items[1]) else full       # flattening rule of the reference
            self.arrays[name] = (k.n_patches * size ** k.dim * leap, leap, size)
        self.primary = k.items[0]
        self.consts = list(k.inputs)
        self.functions = {}               # name -> device function generated from the SymPy body (see _function)
        self.statements = []
        dconst = {}
        for n, (lhs, rhs, direction, span) in enumerate(zip(k.LHS, k.RHS, k.directions, k.struct_inclusion)):
            if str(lhs) in k.directional_consts:
                dconst[str(lhs)] = rhs
                continue
            self.statements.append(self._analyse(n, lhs, rhs, direction, span, dict(dconst)))
        if not self.statements:
            raise LoweringRefused("the statement list is empty")

    # -- analysis ------------------------------------------------------------------------------------------
    def _analyse(self, n, lhs, rhs, direction, span, dconst):
        k = self.k
        where = "statement %d (`%s = %s`)" % (n, lhs, rhs)
        bare = (rhs is None or (isinstance(rhs, str) and rhs == "")) and self._is_call(lhs)      # `Flux(Q, normal, F);`
        calls = []
        for e in (lhs, rhs):
            if isinstance(e, sympy.Basic):
                calls += [a for a in e.atoms(sympy.Function) if self._is_call(a)]
        ptr_width = {}                                        # Indexed argument of a call -> doubles the callee touches from there
        out_ptr = None
        for c in calls:
            f = self._function(c, bare and c == lhs, where, dconst)
            for a, kind, width in zip(c.args, f["kinds"], f["widths"]):
                if kind in ("in", "out"):
                    ptr_width[a] = max(ptr_width.get(a, 1), width)
                if kind == "out":
                    out_ptr = a
        if bare:
            if out_ptr is None:
                raise LoweringRefused("%s: a bare call needs an array out-parameter (its last array argument)" % where)
        elif not isinstance(lhs, Indexed):
            raise LoweringRefused("%s: the left-hand side is not an array element" % where)
        if not bare and (rhs is None or (isinstance(rhs, str) and rhs == "")):
            raise LoweringRefused("%s has no right-hand side" % where)
        target = out_ptr if bare else lhs
        rng = self.cpp._ranges([lhs, rhs], direction, span)
        names = [str(i) for i in k.indexes]
        bounds = {nm: (lo, hi - 1) for nm, (lo, hi) in zip(names, rng)}
        var_loop = rng[-1][1] > 1
        # bounds of every access, dependence of the written array
        reads = set()
        for e in ((lhs, rhs) if not bare else (lhs,)):
            if isinstance(e, sympy.Basic):
                reads |= e.atoms(Indexed)
        accesses = [target] + sorted((a for a in reads if a is not target), key=str)
        for a in accesses:
            base = str(a.base)
            if base not in self.arrays:
                raise LoweringRefused("%s indexes `%s`, which is not a declared item" % (where, base))
            total, leap, size = self.arrays[base]
            strides = [leap * size ** (k.dim - m) for m in range(k.dim + 1)]           # patch, axes...
            lo = hi = 0
            for m, ix in enumerate(a.indices):
                syms = [s for s in ix.free_symbols]
                if len(syms) > 1 or any(str(s) not in bounds for s in syms):
                    raise LoweringRefused("%s: index `%s` of `%s` is not a loop index plus an offset" % (where, ix, base))
                if str(ix) == "var" or (syms and str(syms[0]) == "var"):
                    vlo, vhi = (bounds["var"] if var_loop else (0, 0))
                    lo, hi = lo + vlo, hi + vhi
                    continue
                stride = strides[m] if m < len(strides) else 1
                if syms:
                    s0 = syms[0]
                    coeff = ix.coeff(s0)
                    if coeff != 1:
                        raise LoweringRefused("%s: index `%s` of `%s` is not a loop index plus an offset" % (where, ix, base))
                    b = bounds[str(s0)]
                    off = int(ix - s0)
                    lo, hi = lo + stride * (b[0] + off), hi + stride * (b[1] + off)
                else:
                    lo, hi = lo + stride * int(ix), hi + stride * int(ix)
            hi += ptr_width.get(a, 1) - 1                     # an array argument of a call: the callee touches `width` doubles from there
            if lo < 0 or hi >= total:
                raise LoweringRefused("%s: `%s` reaches flat index %d .. %d of an array of %d doubles over the loop range -- out of "
                                      "bounds (the reference's text would read or write past its heap block)" % (where, a, lo, hi, total))
            if a is not target and str(a.base) == str(target.base) and (tuple(a.indices) != tuple(target.indices) or a in ptr_width or target in ptr_width):
                raise LoweringRefused("%s writes `%s` and reads it at another index (`%s`): a loop-carried dependence in the reference's "
                                      "sequential loop nest, not reproducible by a parallel launch" % (where, target.base, a))
        if bare and var_loop:
            raise LoweringRefused("%s: a bare call inside a loop over the variables would run once per variable" % where)
        for a, w in ptr_width.items():
            if a is not out_ptr and str(a.base) == str(target.base) and (var_loop or tuple(a.indices) != tuple(target.indices)):
                raise LoweringRefused("%s writes `%s` and hands it to a call by address (`&%s`, %d doubles from there): the lanes of the variable "
                                      "loop write what the callee reads -- a loop-carried dependence in the reference's sequential loop nest, not "
                                      "reproducible by a parallel launch" % (where, target.base, a, w))
        if calls:
            ct = _CallText(self.cpp, k.functions)
            text_l, text_r = ct.doprint(lhs), ("" if bare else ct.doprint(rhs))
        else:
            text_l, text_r = self.cpp.Cppify(lhs), self.cpp.Cppify(_ctext(rhs))
        if not var_loop:
            text_l, text_r = text_l.replace(" + var", ""), text_r.replace(" + var", "")
        if "**" in text_r or re.search(r"(?<![\w.])\d+/\d+(?![\w.])", text_r):
            raise LoweringRefused("%s prints as `%s`: not the same arithmetic in C++ (a power or an integer quotient)" % (where, text_r))
        free = set()
        for e in (lhs, rhs):
            if isinstance(e, sympy.Basic):
                free |= {str(s) for s in e.free_symbols if isinstance(s, Symbol)}
        unknown = free - set(names) - set(self.consts) - set(dconst) - {"dim", "patch_size", "halo_size", "n_real", "n_aux"} - set(k.functions)
        unknown = {u for u in unknown if u not in self.arrays}
        if unknown:
            raise LoweringRefused("%s uses %s, which is neither an input constant, a directional constant nor a builder constant"
                                  % (where, sorted(unknown)))
        return {"n": n, "lhs": text_l, "rhs": text_r, "ranges": rng, "var_loop": var_loop, "bare_call": bare,
                "dconst": {nm: v for nm, v in dconst.items() if nm in free}, "writes": str(target.base)}

    # -- functions with SymPy bodies ---------------------------------------------------------------------
    def _is_call(self, e):
        return isinstance(e, sympy.Function) and type(e).__name__ in self.k.functions

    def _function(self, call, is_bare, where, dconst):
        """The `__device__` function behind a call, generated once per name from the SymPy body: parameter kinds from the call site (array
        argument -> pointer; directional constant -> integer, the body is evaluated once per value; anything else -> double), the body's
        expressions as C text with shared sub-expressions named."""
        from ..pde_codegen import _DevicePrinter
        k = self.k
        name = type(call).__name__
        body = k.function_bodies.get(name) or (_default_max if name == "max" else None)
        if body is None:
            raise LoweringRefused("%s calls the opaque function `%s`: the reference resolves it to host C++ at link time; give it a SymPy body "
                                  "(kernel.function(..., body=...)) or use a recognised scheme with pde=" % (where, name))
        kinds, widths = [], []
        idx_args = [j for j, a in enumerate(call.args) if isinstance(a, Indexed)]
        for j, a in enumerate(call.args):
            if isinstance(a, Indexed):
                base = str(a.base)
                if base not in self.arrays:
                    raise LoweringRefused("%s: `%s` is not a declared item" % (where, base))
                leap = self.arrays[base][1]
                kinds.append("out" if (is_bare and j == idx_args[-1]) else "in")
                widths.append(min(leap, k.n_real))
            elif isinstance(a, Symbol) and str(a) in k.directional_consts:
                kinds.append("dir")
                widths.append(0)
            elif isinstance(a, sympy.Basic) and not a.atoms(Indexed):
                kinds.append("num")
                widths.append(0)
            else:
                raise LoweringRefused("%s: argument `%s` of `%s` is neither an array element, a directional constant nor a scalar expression" % (where, a, name))
        if name in self.functions:
            f = self.functions[name]
            if f["kinds"] != kinds or (f["widths"] != widths and "out" not in kinds):
                raise LoweringRefused("%s: `%s` is called with another argument pattern than before" % (where, name))
            return f
        dir_names = [str(a) for a, kd in zip(call.args, kinds) if kd == "dir"]
        if len(dir_names) > 1:
            raise LoweringRefused("%s: more than one directional constant in a call of `%s`" % (where, name))
        values = [int(v) for v in k.directional_consts[dir_names[0]]] if dir_names else [None]
        params, syms = [], []
        for j, (kd, w) in enumerate(zip(kinds, widths)):
            if kd == "in":
                params.append("double* p%d" % j)                    # (non-const, as the reference declares them: `Unit test/Functions.h:2-4`; and an exact match beats the library templates of the same name, e.g. max)
                syms.append([Symbol("p%d_%d" % (j, v), real=True) for v in range(w)])
            elif kd == "out":
                params.append("double* o%d" % j)
                syms.append(None)
            elif kd == "dir":
                params.append("int n%d" % j)
                syms.append("dir")
            else:
                params.append("double s%d" % j)
                syms.append(Symbol("s%d" % j, real=True))
        pr = _DevicePrinter()
        rename = {}
        for j, sy in enumerate(syms):
            if isinstance(sy, list):
                rename.update({x: Symbol("p%d[%d]" % (j, v)) for v, x in enumerate(sy)})
        cases, n_out = {}, None
        for val in values:
            args = [(val if sy == "dir" else sy) for sy in syms if sy is not None]
            try:
                res = body(*args)
            except TypeError as e:
                raise LoweringRefused("%s: the body of `%s` does not take the call's arguments (%s)" % (where, name, e))
            is_list = isinstance(res, (list, tuple))
            if is_list != ("out" in kinds):
                raise LoweringRefused("%s: `%s` %s" % (where, name, "returns a list of expressions: call it as a bare statement with an array out-parameter"
                                                       if is_list else "returns one expression but is called as a bare statement"))
            exprs = [sympy.sympify(e).xreplace(rename) for e in (res if is_list else [res])]
            repl, red = sympy.cse(exprs, symbols=sympy.numbered_symbols("t_"))
            cases[val] = {"temps": [(str(a), pr.doprint(b)) for a, b in repl], "results": [pr.doprint(e) for e in red]}
            n_out = len(exprs)
        if "out" in kinds:
            jo = kinds.index("out")
            room = self.arrays[str(call.args[jo].base)][1]
            if n_out > room:
                raise LoweringRefused("%s: the body of `%s` returns %d expressions, its out-parameter `%s` holds %d entries per volume"
                                      % (where, name, n_out, call.args[jo].base, room))
            widths[jo] = n_out
        f = {"name": name, "kinds": kinds, "widths": widths, "params": params, "cases": cases, "returns": "void" if "out" in kinds else "double",
             "dir_param": next((p.split()[-1] for p, kd in zip(params, kinds) if kd == "dir"), None),
             "out_param": next((p.split()[-1] for p, kd in zip(params, kinds) if kd == "out"), None)}
        self.functions[name] = f
        return f

    def functions_source(self):
        out = []
        for f in self.functions.values():
            out.append("static __device__ inline %s %s(%s) {" % (f["returns"], f["name"], ", ".join(f["params"])))

            def block(c, ind):
                lines = ["%sconst double %s = %s;" % (ind, a, b) for a, b in c["temps"]]
                if f["returns"] == "void":
                    lines += ["%s%s[%d] = %s;" % (ind, f["out_param"], v, e) for v, e in enumerate(c["results"])]
                else:
                    lines.append("%sreturn %s;" % (ind, c["results"][0]))
                return lines
            if f["dir_param"] is None:
                out += block(f["cases"][None], "    ")
            else:
                out.append("    switch (%s) {" % f["dir_param"])
                for val, c in f["cases"].items():
                    out += ["    case %d: {" % val] + block(c, "        ") + ["    } break;" if f["returns"] == "void" else "    }"]
                out.append("    }")
                if f["returns"] == "double":
                    out.append("    return 0.0;")
            out += ["}", ""]
        return out

    # -- code ----------------------------------------------------------------------------------------------
    def array_order(self):
        """Kernel-argument order of the arrays: the primary (caller-owned) first, then the temporaries by name."""
        return [self.primary] + sorted(a for a in self.arrays if a != self.primary)

    def source(self):
        k = self.k
        names = [str(i) for i in k.indexes]
        arrs = self.array_order()
        params = ", ".join(["double* %s" % a for a in arrs] + ["double %s" % c for c in self.consts])
        out = ["// Generated by exahype_amd.printers.lowering for `%s`: one kernel per statement of the builder state, the reference's loop"
               % self.name,
               "// ranges, AoS strides and expression text (exahype/printers/CPPPrinter.py:84-137); gfx950, -ffp-contract=off.",
               "#include <hip/hip_runtime.h>", "#include <cmath>", ""]
        out += self.functions_source()
        lits = "".join("    const %s\n" % lit for lit in k.literals)
        for s in self.statements:
            rng = s["ranges"]
            ext = [hi - lo for lo, hi in rng]
            if not s["var_loop"]:
                ext[-1] = 1
            total = 1
            for e in ext:
                total *= e
            body = ["__global__ void __launch_bounds__(256) stmt_%d(%s) {" % (s["n"], params),
                    "    const long t_ = (long)blockIdx.x * 256 + threadIdx.x;",
                    "    if (t_ >= %dL) return;" % total, "    long r_ = t_;"]
            for nm, (lo, hi), e in reversed(list(zip(names, rng, ext))):      # innermost (var) fastest
                if nm == "var" and not s["var_loop"]:
                    continue
                body.append("    const int %s = %d + (int)(r_ %% %d); r_ /= %d;" % (nm, lo, e, e))
            if lits:
                body.append(lits.rstrip("\n"))
            for nm, v in s["dconst"].items():
                body.append("    const double %s = %s;" % (nm, v))
            body.append(("    %s;" % s["lhs"]) if s.get("bare_call") else ("    %s = %s;" % (s["lhs"], s["rhs"])))
            body.append("}")
            out += body + [""]
            s["total"] = total
        out.append('extern "C" int exa_generated_%s(double** arrays_, const double* consts_, void* stream_) {' % self.name)
        out.append("    hipStream_t s_ = static_cast<hipStream_t>(stream_);")
        call = ", ".join(["arrays_[%d]" % i for i in range(len(arrs))] + ["consts_[%d]" % i for i in range(len(self.consts))])
        for s in self.statements:
            out.append("    hipLaunchKernelGGL(stmt_%d, dim3(%d), dim3(256), 0, s_, %s);" % (s["n"], (s["total"] + 255) // 256, call))
        out += ["    return hipGetLastError() == hipSuccess ? 0 : -1;", "}", ""]
        return "\n".join(out)

    def key(self):
        return hashlib.sha256(self.source().encode()).hexdigest()[:16]

    def build(self, force=False):
        """hipcc (gfx950, no FMA contraction: the reference's text evaluates every product and sum separately) -> side library."""
        from .. import build as _build
        d = os.path.join(GEN_DIR, "gen_" + self.key())
        so = os.path.join(d, "libexahype_generated.so")
        if os.path.exists(so) and not force:
            return so
        with _build.build_lock(d):
            if os.path.exists(so) and not force:          # another process built it meanwhile (renamed into place: complete)
                return so
            src = os.path.join(d, "generated.hip")
            with open(src, "w") as f:
                f.write(self.source())
            hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
            _build.link_atomically([hipcc, "-O3", "-fPIC", "-shared", "-std=c++17", "--offload-arch=%s" % _build.ARCH, "-ffp-contract=off",
                                    "-Wno-unused-variable", "-Wno-unused-value", src], so)
        return so

    # -- execution -----------------------------------------------------------------------------------------
    def bind(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("the generated kernels need a GPU (there is no CPU path)")
        self._lib = C.CDLL(self.build())
        self._fn = getattr(self._lib, "exa_generated_%s" % self.name)
        self._fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.c_void_p]
        self._fn.restype = C.c_int
        self._dev = torch.device("cuda", device)
        # temporaries: allocated once per bound kernel (the reference allocates and frees them inside every call; their
        # content before a statement writes them is unspecified there -- heap garbage -- and zero here)
        self._tmp = {a: torch.zeros(self.arrays[a][0], dtype=torch.float64, device=self._dev) for a in self.array_order()[1:]}
        return self

    def run(self, Q, *consts):
        """`time_step(Q, consts...)`: Q = the primary item, a numpy array (staged through the device, updated in place) or a CUDA
        tensor (resident); consts in the order of the builder's `const()` inputs."""
        import numpy as np
        import torch
        if getattr(self, "_fn", None) is None:
            self.bind()
        if len(consts) != len(self.consts):
            raise TypeError("%s takes %d constant(s) %s, got %d" % (self.name, len(self.consts), self.consts, len(consts)))
        need = self.arrays[self.primary][0]
        host = None
        if isinstance(Q, np.ndarray):
            if Q.dtype != np.float64 or Q.size != need or not Q.flags.c_contiguous:
                raise ValueError("`%s` must be a contiguous float64 array of %d doubles" % (self.primary, need))
            host, q = Q, torch.as_tensor(Q.reshape(-1), device=self._dev)
        else:
            q = Q
            if not (q.is_cuda and q.dtype == torch.float64 and q.numel() == need and q.is_contiguous()):
                raise ValueError("`%s` must be a contiguous float64 CUDA tensor of %d doubles" % (self.primary, need))
        ptrs = (C.c_void_p * len(self.arrays))(*([q.data_ptr()] + [self._tmp[a].data_ptr() for a in self.array_order()[1:]]))
        cs = (C.c_double * max(1, len(consts)))(*[float(c) for c in consts])
        stream = torch.cuda.current_stream(self._dev).cuda_stream
        if self._fn(ptrs, cs, C.c_void_p(stream)) != 0:
            raise RuntimeError("launch of the generated kernels failed")
        if host is not None:
            host.reshape(-1)[:] = q.cpu().numpy()
        return Q
