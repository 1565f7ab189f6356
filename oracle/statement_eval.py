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


def run_statements(statements, index_names, arrays, consts, literals=None):
    """statements: dicts with `lhs`, `rhs` (C text), `ranges` ([lo, hi) per index name, var last), `var_loop`, `dconst`.
    arrays: name -> flat float64 numpy array (updated in place); consts: name -> float."""
    env_base = dict(literals or {})
    env_base.update(consts)
    env_base.update({"sqrt": np.sqrt, "fabs": np.abs, "abs": np.abs})
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
        m = re.fullmatch(r"\s*(\w+)\[(.*)\]\s*", s["lhs"])
        assert m, s["lhs"]
        target, index_text = m.group(1), m.group(2)
        idx = eval(index_text, {"__builtins__": {}}, env)              # noqa: S307 -- test infrastructure, own text
        val = eval(s["rhs"], {"__builtins__": {}}, env)                # noqa: S307
        arrays[target][idx] = val
    return arrays
