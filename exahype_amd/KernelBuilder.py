"""KernelBuilder -- the operator surface of the reference
(`exahype/KernelBuilder.py:41-227`) for the MI355X back-end.

A user describes a patch kernel as a list of SymPy statements written with 1-D
offsets relative to "the current direction" (`Q[0]`, `Q[-1]`, `Q[1]`); the
builder expands each into full `[patch, i, j, (k), var]` indices, replicates
directional statements per dimension and records, per statement, its direction
and how much of the variable struct it spans.  Printers consume that state
(`inputs, items, directional_items, directional_consts, functions, item_struct,
parents, literals, LHS, RHS, directions, struct_inclusion`) -- the reference's
CPPPrinter to emit C++ text, ours (printers/HIPPrinter.py) to dispatch into
hand-written HIP kernels.

Signatures, defaults, return values, exception types/messages and the recorded
state match the reference (pinned by tests/golden/builder_state_*.json, captured
from the reference itself).  The implementation differs: index expansion is a
SymPy printer over the expression tree rather than character surgery on its
string form.  Two reference behaviours are kept on purpose because printers
depend on them: struct inference matches item names as
substrings of the statement text, and `items[1]` is the halo-less array whose
spatial indices are shifted by -1.
"""
from __future__ import annotations

from typing import List

from sympy import Idx, Indexed, IndexedBase, core, symbols, sympify
from sympy.codegen.ast import none
from sympy.printing.str import StrPrinter

from .TypedFunction import TypedFunction

_DIR_SUFFIX = ['_patch', '_x', '_y', '_z']      # suffix of a directional item per direction (1-based axes)


def viable(dim: int, patch_size: int, halo_size: int):
    """Reference rule (`exahype/KernelBuilder.py:41-48`): 2-D or 3-D, >= 1 volume, halo >= 0."""
    return dim in (2, 3) and patch_size >= 1 and halo_size >= 0


class _IndexExpander(StrPrinter):
    """Prints an expression with every `name[offset]` expanded to the full index tuple."""

    def __init__(self, builder, direction):
        super().__init__()
        self._b = builder
        self._direction = direction

    def _print(self, expr, **kwargs):
        # Indexed carries its own `_sympystr`, which the printer would prefer over a _print_Indexed
        if isinstance(expr, Indexed):
            return self._expand(expr)
        return super()._print(expr, **kwargs)

    def _expand(self, expr):
        b = self._b
        name = str(expr.base)
        if self._direction >= 0 and name in b.directional_items:
            name += _DIR_SUFFIX[self._direction]
        offset = expr.indices[0]
        shifted = len(b.items) > 1 and name == b.items[1]
        scalar_item = b.item_struct.get(name) == 0
        parts = []
        for level, idx in enumerate(b.indexes):
            if scalar_item and str(idx) == 'var':
                continue
            text = str(idx)
            if level == self._direction and offset != 0:
                text += ('+%s' % offset) if offset > 0 else str(offset)
            elif shifted and str(idx) != 'var':
                text += '-1'
            parts.append(text)
        return '%s[%s]' % (name, ','.join(parts))


class KernelBuilder:
    def __init__(self: KernelBuilder, dim: int, patch_size: int, halo_size: int, n_real: int, n_aux: int, n_patches: int = 1):
        if not viable(dim, patch_size, halo_size):
            raise Exception('check viability of inputs')
        self.dim, self.patch_size, self.halo_size = dim, patch_size, halo_size
        self.n_patches, self.n_real, self.n_aux = n_patches, n_real, n_aux

        names = 'patch i j' + (' k' if dim == 3 else '') + ' var'
        self.indexes = list(symbols(names, cls=Idx))

        self.literals = []              # C++ definition lines
        self.parents = {}               # name -> qualifying parent (object or namespace)
        self.inputs = []
        self.input_types = []
        self.items = []                 # names; items[0] = primary in/out, items[1] = halo-less array
        self.directional_items = []
        self.directional_consts = {}    # name -> value per direction
        self.functions = []
        self.function_bodies = {}             # extension: SymPy bodies of PDE terms (see function())
        self.item_struct = {}           # 0: scalar per volume, 1: n_real, 2: n_real + n_aux

        full = (0, patch_size + 2 * halo_size)
        self.default_shape = [n_patches] + [full for _ in range(dim)]
        self.all_items = {'i': Idx('i', full), 'j': Idx('j', full), 'k': Idx('k', full),
                          'patch': Idx('patch', (0, n_patches)), 'var': Idx('var', (0, n_real + n_aux))}

        self.LHS = []
        self.RHS = []
        self.directions = []            # -1: none, 1..dim: axis, -2: writes an input
        self.struct_inclusion = []      # -1: n/a, 0: none, 1: n_real, 2: n_real + n_aux

        for name, value in (('dim', dim), ('patch_size', patch_size), ('halo_size', halo_size), ('n_real', n_real), ('n_aux', n_aux)):
            self.const(name, define=f'int {name} = {value};')

    # -- declarations ----------------------------------------------------------------------
    def const(self: KernelBuilder, expr: str, in_type: str = "double", parent: core.basic.Basic = None, define=None):
        self.all_items[expr] = symbols(expr)
        if parent != None:  # noqa: E711  (SymPy objects overload ==)
            self.parents[expr] = str(parent)
        elif define != None:  # noqa: E711
            self.literals.append(define)
        else:
            self.inputs.append(expr)
            self.input_types.append(in_type)
            return symbols(expr, real=True)
        return symbols(expr)

    def directional_const(self: KernelBuilder, expr: str, vals: List):
        if len(vals) != self.dim:
            raise Exception("directional constant must have values for each direction")
        self.directional_consts[expr] = vals
        self.all_items[expr] = symbols(expr, real=True)
        return symbols(expr, real=True)

    def item(self: KernelBuilder, expr: str, struct: bool = True, in_type: str = "double*", parent=None):
        self.items.append(expr)
        self.all_items[expr] = IndexedBase(expr, real=True)
        if len(self.items) == 1:
            self.input_types.append(in_type)
        self.item_struct[expr] = 2 if struct else 0
        if parent != None:  # noqa: E711
            self.parents[expr] = str(parent)
        return IndexedBase(expr, real=True)

    def directional_item(self: KernelBuilder, expr: str, struct: bool = True):
        self.directional_items.append(expr)
        self.item_struct[expr] = 1 if struct else 0
        for suffix in _DIR_SUFFIX[1:self.dim + 1]:
            self.all_items[expr + suffix] = IndexedBase(expr + suffix, real=True)
            self.item_struct[expr + suffix] = 1 if struct else 0
        return IndexedBase(expr, real=True)

    def function(self: KernelBuilder, expr: str, parent: core.basic.Basic = None, parameter_types: List = [], return_type=none,
                 body=None):
        """As the reference (`exahype/KernelBuilder.py:134`).  Extension: `body`, a callable
        `(q, normal) -> SymPy expression(s)` in the state symbols q -- for the flux a list of n_real expressions, for
        the eigenvalue one expression.  In the reference these names resolve to host C++ at link time
        (`Unit test/Functions.h:2-4`); a device kernel needs the terms themselves, and HIPPrinter compiles the bodies
        of the functions named Flux/flux and maxEigenvalue when both are given.  The builder state the reference's
        printers see is unchanged."""
        if parent != None:  # noqa: E711
            self.parents[expr] = str(parent)
        self.functions.append(expr)
        func = TypedFunction(expr)
        func.returnType(return_type)
        func.parameterTypes(parameter_types)
        self.all_items[expr] = func
        if body is not None:
            if not callable(body):
                raise TypeError("body must be callable: (q, normal) -> SymPy expression(s)")
            self.function_bodies[expr] = body
        return func

    # -- statements ------------------------------------------------------------------------
    def single(self: KernelBuilder, LHS: core.basic.Basic, RHS: core.basic.Basic = None, direction: int = -1, struct: bool = False):
        writes_input = str(LHS).partition('[')[0] in self.inputs
        if struct:
            span = 1
        elif str(type(LHS)) in self.functions or str(type(RHS)) in self.functions:
            span = 0
        elif writes_input:
            span = 2
        else:
            text = str(LHS) + str(RHS)
            span = min(v for name, v in self.item_struct.items() if name in text)
        self.struct_inclusion.append(span)
        self.directions.append(-2 if writes_input else direction)
        self.LHS.append(self.index(LHS, direction))
        self.RHS.append(self.index(RHS, direction))

    def directional(self: KernelBuilder, LHS: core.basic.Basic, RHS: core.basic.Basic = None, struct: bool = False):
        text = (str(LHS), str(RHS))
        for axis in range(self.dim):
            for name, vals in self.directional_consts.items():
                if name in text[0] or name in text[1]:
                    self.LHS.append(self.all_items[name])
                    self.RHS.append(vals[axis])
                    self.struct_inclusion.append(-1)
                    self.directions.append(-1)
            self.single(LHS, RHS, axis + 1, struct)

    def index(self: KernelBuilder, expr_in, direction: int = -1):
        """Expand 1-D offsets to `[patch, i, j, (k), var]` and re-parse against `all_items`."""
        if isinstance(expr_in, str) and expr_in == '':
            return ''
        if not isinstance(expr_in, core.basic.Basic):
            return sympify(str(expr_in), locals=self.all_items)          # None -> None, numbers -> numbers
        return sympify(_IndexExpander(self, direction).doprint(expr_in), locals=self.all_items)
