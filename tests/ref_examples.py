"""Kernels for the operator-surface tests.

* `reference_example(...)` runs the body of one of the reference's OWN example scripts where it lies under
  /root/reference (lines exec'ed in place, nothing copied; same pattern as tests/golden/make_golden.py) against
  the KernelBuilder it is given.  Tests that need it skip where the reference tree is absent (the GPU box).
* `rusanov_patch_update(...)` / `cell_data_patch_update(...)` hold the STATEMENT LISTS of the reference's examples
  (`examples/Batched_stateless.py:9-35`, `examples/kernel-generator.py:6-45`) with other variable names and free
  dim / sizes: the statement list is the input the HIP back-end's recogniser has to match, so it is the same
  scheme by necessity -- the same six statements in the same order, not an independent design.  Their expected
  output comes from the pinned oracle.
* `builder_state(k)` is the dump the golden builder-state fixtures hold (tests/golden/make_golden.py).
"""
import os

import pytest
from sympy.codegen.ast import integer, none, real

REF = "/root/reference"
EXAMPLES = {"batched_stateless": ("examples/Batched_stateless.py", 9, 35),      # ctor .. last statement
            "kernel_generator": ("examples/kernel-generator.py", 6, 45)}


def have_reference():
    return os.path.isdir(os.path.join(REF, "examples"))


def reference_example(name, KernelBuilder, ctor=None, substitutions=()):
    """exec the example's lines [first, last] with `KernelBuilder` injected; returns its `kernel`.
    ctor: replacement text for the constructor arguments (the script's own are dim=2, patch_size=4, ...);
    substitutions: (old, new) text pairs applied to the body (e.g. the per-direction constant for dim = 3)."""
    if not have_reference():
        pytest.skip("reference tree not present (its example scripts are exec'ed where they lie, never copied)")
    path, first, last = EXAMPLES[name]
    lines = open(os.path.join(REF, path)).read().split("\n")[first - 1:last]
    if ctor is not None:
        assert lines[0].startswith("kernel = KernelBuilder(")
        lines[0] = "kernel = KernelBuilder(%s)" % ctor
    src = "\n".join(lines)
    for old, new in substitutions:
        assert old in src, old
        src = src.replace(old, new)
    env = dict(KernelBuilder=KernelBuilder, integer=integer, real=real, none=none)
    exec(src, env)
    return env["kernel"]


def reference_example_3d_p15(KernelBuilder):
    """The 3-D, two-patch variant the fixture builder_state_3d_p15.json was captured for."""
    return reference_example("batched_stateless", KernelBuilder, ctor="dim=3,patch_size=15,halo_size=1,n_real=5,n_aux=0,n_patches=2",
                             substitutions=(("'normal',[0,1]", "'normal',[0,1,2]"),))


def rusanov_patch_update(KernelBuilder, dim=3, patch_size=6, halo_size=1, n_real=5, n_aux=2, n_patches=3,
                         term_names=("Flux", "maxEigenvalue", "max")):
    """First-order Rusanov patch update written against exahype_amd's surface: state array `U` (with halo), working copy
    `W`, per-direction face flux `numflux` and wave speed `speed`, time step `tau`, axis constant `axis`."""
    kb = KernelBuilder(dim, patch_size, halo_size, n_real, n_aux, n_patches=n_patches)
    U, W = kb.item("U"), kb.item("W")
    numflux = kb.directional_item("numflux")
    speed = kb.directional_item("speed", struct=False)
    tau = kb.const("tau")
    axis = kb.directional_const("axis", list(range(dim)))
    f = kb.function(term_names[0], parameter_types=[U, real, U], return_type=integer)
    lam = kb.function(term_names[1], parameter_types=[U, real], return_type=real)
    mx = kb.function(term_names[2], parameter_types=[U, U], return_type=none)
    kb.single(W[0], U[0])                                              # working copy, halo included
    kb.directional(f(W[0], axis, numflux[0]))                          # flux and wave speed per volume and direction
    kb.directional(speed[0], lam(W[0], axis))
    kb.directional(W[0], W[0] + 0.5 * (numflux[-1] - numflux[1]))      # central flux difference
    lo = -mx(speed[-1], speed[0]) * (U[0] - U[-1])                     # dissipation across the low / high face
    hi = -mx(speed[1], speed[0]) * (U[0] - U[1])
    kb.directional(W[0], W[0] + 0.5 * tau * (lo - hi), struct=True)
    kb.single(U[0], W[0])
    return kb


def cell_data_patch_update(KernelBuilder, dim=2, patch_size=4, halo_size=1, n_real=4, n_aux=0, n_patches=1,
                           solver="demo::instanceOfSolver"):
    """The `exahype2::CellData` flavour of the patch update written against exahype_amd's surface with names of its own (the calls the
    reference's examples/kernel-generator.py makes): a CellData object `cell` with the input array `Uin` (with halo) and the halo-less
    output `Uout`, PDE terms `F` / `lambdaMax` as methods of a solver object that take the volume centre, the volume size, t and dt."""
    kb = KernelBuilder(dim=dim, patch_size=patch_size, halo_size=halo_size, n_real=n_real, n_aux=n_aux, n_patches=n_patches)
    cell = kb.item('cell', in_type='::exahype2::CellData&')
    kb.const('stopwatch', in_type='::tarch::timing::Measurement&')
    Uout = kb.item('Uout', parent=cell)
    Uin = kb.item('Uin', parent=cell)
    numflux = kb.directional_item('numflux')
    speed = kb.directional_item('speed', struct=False)
    tau = kb.const('tau', parent=cell)
    now = kb.const('now', parent=cell)
    axis = kb.directional_const('axis', tuple(range(dim)))
    mid = kb.const('mid', parent=cell)
    edge = kb.const('edge', parent=cell)
    F = kb.function('F', parent=solver)
    lam = kb.function('lambdaMax', parent=solver)
    mx = kb.function('max')
    where = kb.function('getVolumeCentre', parent='exahype2::fv::')
    width = kb.function('getVolumeSize', parent='exahype2::fv::')
    P = kb.all_items["patch_size"]
    idx = {kb.all_items["i"], kb.all_items["j"]} if dim == 2 else {kb.all_items["i"], kb.all_items["j"], kb.all_items["k"]}
    kb.single(Uin[0], Uout[0])
    kb.directional(F(Uin[0], where(mid, edge, P, idx), width(edge, P), now, tau, axis, numflux[0]))
    kb.directional(speed[0], F(Uin[0], where(mid, edge, P), width(edge, P), now, tau, axis))
    kb.directional(Uin[0], Uin[0] + 0.5 * (numflux[-1] - numflux[1]))
    lo = -mx(speed[-1], speed[0]) * (Uout[0] - Uout[-1])
    hi = -mx(speed[1], speed[0]) * (Uout[0] - Uout[1])
    kb.directional(Uin[0], Uin[0] + 0.5 * tau * (lo - hi), struct=True)
    kb.single(Uout[0], Uin[0])
    return kb


def builder_state(k):
    """The observable state printers read (same dump as tests/golden/make_golden.py)."""
    return dict(
        dim=k.dim, patch_size=k.patch_size, halo_size=k.halo_size, n_patches=k.n_patches, n_real=k.n_real,
        n_aux=k.n_aux, indexes=[str(i) for i in k.indexes],
        inputs=list(k.inputs), input_types=list(k.input_types), items=list(k.items),
        directional_items=list(k.directional_items),
        directional_consts={a: list(b) for a, b in k.directional_consts.items()},
        functions=list(k.functions), item_struct=dict(k.item_struct), parents=dict(k.parents),
        literals=list(k.literals), all_items=sorted(k.all_items.keys()),
        LHS=[str(x) for x in k.LHS], RHS=[str(x) for x in k.RHS], directions=list(k.directions),
        struct_inclusion=list(k.struct_inclusion),
        function_types={f: dict(return_type=str(k.all_items[f].returnType()),
                                parameter_types=[str(p) for p in (k.all_items[f].parameterTypes() or [])])
                        for f in k.functions},
    )
