"""The reference's two example kernels, written against the operator surface exactly as a user
script would (`examples/Batched_stateless.py:9-35`, `examples/kernel-generator.py:6-45`), plus a
parametrised variant.  Used by the surface tests: the same calls must produce the state captured
from the reference (tests/golden/builder_state_*.json)."""
from sympy.codegen.ast import integer, none, real


def batched_stateless(KernelBuilder, dim=2, patch_size=4, halo_size=1, n_real=5, n_aux=5, n_patches=1):
    kernel = KernelBuilder(dim=dim, patch_size=patch_size, halo_size=halo_size, n_real=n_real, n_aux=n_aux, n_patches=n_patches)
    Q = kernel.item('Q')
    Q_copy = kernel.item('Q_copy')
    tmp_flux = kernel.directional_item('tmp_flux')
    tmp_eig = kernel.directional_item('tmp_eigen', struct=False)
    dt = kernel.const('dt')
    normal = kernel.directional_const('normal', list(range(dim)))
    Flux = kernel.function('Flux', parameter_types=[Q, real, Q], return_type=integer)
    Eigen = kernel.function('maxEigenvalue', parameter_types=[Q, real], return_type=real)
    Max = kernel.function('max', parameter_types=[Q, Q], return_type=none)
    kernel.single(Q_copy[0], Q[0])
    kernel.directional(Flux(Q_copy[0], normal, tmp_flux[0]))
    kernel.directional(tmp_eig[0], Eigen(Q_copy[0], normal))
    kernel.directional(Q_copy[0], Q_copy[0] + 0.5 * (tmp_flux[-1] - tmp_flux[1]))
    left = -Max(tmp_eig[-1], tmp_eig[0]) * (Q[0] - Q[-1])
    right = -Max(tmp_eig[1], tmp_eig[0]) * (Q[0] - Q[1])
    kernel.directional(Q_copy[0], Q_copy[0] + 0.5 * dt * (left - right), struct=True)
    kernel.single(Q[0], Q_copy[0])
    return kernel


def kernel_generator(KernelBuilder):
    kernel = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=4, n_aux=0)
    Data = kernel.item('patchData', in_type='::exahype2::CellData&')
    kernel.const('timingComputeKernel', in_type='::tarch::timing::Measurement&')
    Q = kernel.item('QOut', parent=Data)
    Q_copy = kernel.item('QIn', parent=Data)
    tmp_flux = kernel.directional_item('tmp_flx')
    tmp_eig = kernel.directional_item('tmp_eigen', struct=False)
    dt = kernel.const('dt', parent=Data)
    t = kernel.const('t', parent=Data)
    normal = kernel.directional_const('normal', (0, 1))
    cellCentre = kernel.const('cellCentre', parent=Data)
    cellSize = kernel.const('cellSize', parent=Data)
    solver = 'benchmarks::exahype2::kernelbenchmarks::repositories::instanceOfFVRusanovSolver'
    Flux = kernel.function('flux', parent=solver)
    kernel.function('maxEigenvalue', parent=solver)
    Max = kernel.function('max')
    Centre = kernel.function('getVolumeCentre', parent='exahype2::fv::')
    Size = kernel.function('getVolumeSize', parent='exahype2::fv::')
    patch_size = kernel.all_items["patch_size"]
    i = kernel.all_items["i"]
    j = kernel.all_items["j"]
    kernel.single(Q_copy[0], Q[0])
    kernel.directional(Flux(Q_copy[0], Centre(cellCentre, cellSize, patch_size, {i, j}), Size(cellSize, patch_size), t, dt, normal, tmp_flux[0]))
    kernel.directional(tmp_eig[0], Flux(Q_copy[0], Centre(cellCentre, cellSize, patch_size), Size(cellSize, patch_size), t, dt, normal))
    kernel.directional(Q_copy[0], Q_copy[0] + 0.5 * (tmp_flux[-1] - tmp_flux[1]))
    left = -Max(tmp_eig[-1], tmp_eig[0]) * (Q[0] - Q[-1])
    right = -Max(tmp_eig[1], tmp_eig[0]) * (Q[0] - Q[1])
    kernel.directional(Q_copy[0], Q_copy[0] + 0.5 * dt * (left - right), struct=True)
    kernel.single(Q[0], Q_copy[0])
    return kernel


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
