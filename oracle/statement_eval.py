"""TEST INFRASTRUCTURE (not shipped, never imported by the product): CPU evaluation of a lowered statement list.

The reference's printer (`exahype/printers/CPPPrinter.py:84-137`) emits, per statement, one loop nest `lhs = rhs;` over index
ranges; with no loop-carried dependence (the lowering refuses those) the nest is one vectorised assignment.  This evaluator takes
the statements in the form the C++ text has them -- flat-index expressions such as `Q[108*patch + 18*i + 3*j + var]` -- and
evaluates them with numpy on index grids: Python's operator precedence and left-to-right association are C's, numpy's float64
operations are IEEE without contraction, so the result is what the reference's loop nest computes, operation for operation.
Parity of the ranges / strides / text themselves is pinned elsewhere (tests/golden/cppprinter_*.txt against the reference's output).
"""
import re

import numpy as np


class _Ptr:
    """`&A[index]` as the callee sees it: p[v] is A[index + v] (index: an integer grid, one entry per loop iteration)"""

    def __init__(self, arr, idx):
        self.arr, self.idx = arr, idx

    def __getitem__(self, v):
        return self.arr[self.idx + v]

    def __setitem__(self, v, val):
        self.arr[self.idx + v] = val


_MATH = {"sqrt": np.sqrt, "fabs": np.abs, "abs": np.abs, "fmax": np.fmax, "fmin": np.fmin, "pow": np.power, "exp": np.exp, "log": np.log,
         "sin": np.sin, "cos": np.cos}


def device_functions(func_defs):
    """Python callables for the `__device__` functions the lowering generated from SymPy bodies (StatementLowering.functions): the SAME C
    text (named sub-expressions, then the results), evaluated line by line with numpy -- so a call in a statement computes, operation for
    operation, what the generated function computes."""
    out = {}
    for name, f in func_defs.items():
        pnames = [p.split()[-1].lstrip("*") for p in f["params"]]

        def call(*args, _f=f, _pn=pnames):
            env = dict(_MATH)
            env.update(dict(zip(_pn, args)))
            key = None if _f["dir_param"] is None else int(np.asarray(env[_f["dir_param"]]).reshape(-1)[0])
            c = _f["cases"][key]
            for tname, text in c["temps"]:
                env[tname] = eval(text, {"__builtins__": {}}, env)                   # noqa: S307 -- test infrastructure, own text
            vals = [eval(text, {"__builtins__": {}}, env) for text in c["results"]]    # noqa: S307
            if _f["returns"] == "void":
                o = env[_f["out_param"]]
                for v, val in enumerate(vals):
                    o[v] = val
                return None
            return vals[0]
        out[name] = call
    return out


def run_statements(statements, index_names, arrays, consts, literals=None, functions=None):
    """statements: dicts with `lhs`, `rhs` (C text), `ranges` ([lo, hi) per index name, var last), `var_loop`, `dconst`, `bare_call`.
    arrays: name -> flat float64 numpy array (updated in place); consts: name -> float; functions: StatementLowering.functions."""
    env_base = dict(literals or {})
    env_base.update(consts)
    env_base.update(_MATH)
    env_base.update(device_functions(functions or {}))
    env_base["_P"] = _Ptr
    addr = lambda text: re.sub(r"&(\w+)\[([^\]]*)\]", r"_P(\1, \2)", text)          # `&A[index]` -> _P(A, index)
    for s in statements:
        names = list(index_names)
        rng = list(s["ranges"])
        if not s["var_loop"]:
            names, rng = names[:-1], rng[:-1]
        grids = np.meshgrid(*[np.arange(lo, hi) for lo, hi in rng], indexing="ij")
        env = dict(env_base)
        env.update({n: g for n, g in zip(names, grids)})
        env.update(s.get("dconst", {}))
        env.update(arrays)
        if s.get("bare_call"):                                         # `Flux(&Q[...], normal, &F[...]);` -- the callee writes through its out-parameter
            eval(addr(s["lhs"]), {"__builtins__": {}}, env)            # noqa: S307 -- test infrastructure, own text
            continue
        m = re.fullmatch(r"\s*(\w+)\[(.*)\]\s*", s["lhs"])
        assert m, s["lhs"]
        target, index_text = m.group(1), m.group(2)
        idx = eval(index_text, {"__builtins__": {}}, env)              # noqa: S307 -- test infrastructure, own text
        val = eval(addr(s["rhs"]), {"__builtins__": {}}, env)          # noqa: S307
        arrays[target][idx] = val
    return arrays
