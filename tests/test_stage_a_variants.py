"""GPU parity of the two stage-A kernels built for 3-D, N = 6 (include/exahype_hip.h EXA_STAGE_A_LDS / EXA_STAGE_A_REG):
each against the CPU oracle on the same seeded inputs (relative 1e-10, the north-star's tolerance; ADER-DG is "parity
unpinned" against the reference, which holds no ADER-DG -- see tests/test_gpu_parity.py), and against each other.

The register-resident kernel (exa_dg_reg.hpp) runs a persistent grid of two workgroups per CU that walk over the cells, so
the cases cover: fewer cells than workgroups, more cells than resident workgroups (several cells per workgroup, not a
multiple), sub-boxes of a block (shell / interior launches of the sharded step), anisotropic cells, 1..N Picard iterations.
"""
import numpy as np
import pytest

from tests.util import euler_dg_state, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10
N = 6


@pytest.fixture(scope="module")
def exa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from exahype_amd import solvers
    return solvers


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.lib()
    return oracle


def _ops():
    from oracle.dg_operators import operators
    return operators(N)


@pytest.mark.parametrize("variant", ["reg", "lds"])
@pytest.mark.parametrize("nc,n_it", [((2, 2, 2), -1), ((1, 1, 1), -1), ((3, 2, 1), 1), ((2, 1, 2), 2), ((2, 3, 2), 3), ((12, 10, 9), -1)])
def test_stage_a_variant_vs_oracle(exa, orc, variant, nc, n_it):
    ops = _ops()
    u = euler_dg_state(tuple(nc) + (N,) * 3, seed=9000 + sum(nc) + max(n_it, 0))
    dx = [1.0 / nc[0], 0.8 / nc[1], 1.3 / nc[2]]                       # anisotropic: the per-direction scale is a lane property in the reg kernel
    dt = 0.02 * min(dx) / (2 * N - 1)
    nit = N if n_it < 0 else n_it
    s = exa.AderDgSolver(3, N, nc, n_picard=n_it, dx=dx, stage_a=variant)
    s.upload(u)
    s.predictor_volume(dt)
    us_o, tr_o = orc.aderdg_stage_a(u.reshape(-1), dt, dx, ops, 3, N, 5, orc.PDE_EULER, nit)
    assert rel_err(s.download().reshape(-1), us_o) < TOL
    assert rel_err(s.trace.cpu().numpy().reshape(tr_o.shape), tr_o) < TOL
    # two full steps (stage B reads what stage A left)
    s.upload(u)
    uo = u.reshape(-1).copy()
    for _ in range(2):
        s.step(dt)
        uo = orc.aderdg_step(uo, dt, dx, ops, 3, N, 5, orc.PDE_EULER, nit, nc)
    assert rel_err(s.download().reshape(-1), uo) < TOL


@pytest.mark.parametrize("variant", ["reg", "lds"])
@pytest.mark.parametrize("nc,n_it", [((1, 2, 2), -1), ((2, 1, 1), 1), ((2, 2, 1), 3), ((7, 7, 6), -1)])
def test_stage_a_variant_n8_vs_oracle(exa, orc, variant, nc, n_it):
    """cfg 4's order (N = 8): "reg" = the matrix-pipe kernel with the iterate in registers (exa_dg_m8.hpp), "lds" = the level-streamed
    kernel with the slab (exa_dg_stream.hpp); 294 cells > 256 resident workgroups in the last case."""
    from oracle.dg_operators import operators
    N8 = 8
    ops = operators(N8)
    u = euler_dg_state(tuple(nc) + (N8,) * 3, seed=800 + sum(nc))
    dx = [1.0 / nc[0], 0.9 / nc[1], 1.2 / nc[2]]
    dt = 0.02 * min(dx) / (2 * N8 - 1)
    nit = N8 if n_it < 0 else n_it
    s = exa.AderDgSolver(3, N8, nc, n_picard=n_it, dx=dx, stage_a=variant)
    assert ("m8" in s.stage_a_kernel_name()) == (variant == "reg")
    s.upload(u)
    s.predictor_volume(dt)
    us_o, tr_o = orc.aderdg_stage_a(u.reshape(-1), dt, dx, ops, 3, N8, 5, orc.PDE_EULER, nit)
    assert rel_err(s.download().reshape(-1), us_o) < TOL
    assert rel_err(s.trace.cpu().numpy().reshape(tr_o.shape), tr_o) < TOL


def test_stage_a_variants_agree_on_boxes(exa):
    """shell / interior box launches of the 2x2x2 partition: both kernels, box by box, give the same block (to rounding)."""
    nc = (9, 8, 7)
    u = euler_dg_state(tuple(nc) + (N,) * 3, seed=4242)
    dt = 1e-3
    out = {}
    for variant in ("reg", "lds"):
        s = exa.AderDgSolver(3, N, nc, stage_a=variant)
        s.upload(u)
        shell, interior = exa.CartesianPartition(8, 0, 3).shell_and_interior(nc)
        for lo, hi in shell + [interior]:
            s.predictor_volume(dt, lo, hi)
        out[variant] = (s.download().copy(), s.trace.cpu().numpy().copy())
    assert rel_err(out["reg"][0], out["lds"][0]) < 1e-12
    assert rel_err(out["reg"][1], out["lds"][1]) < 1e-12
    assert not np.array_equal(out["reg"][0], u)


def test_stage_a_variant_rejects_unknown(exa):
    from exahype_amd._lib import ExaHypeHipError
    s = exa.AderDgSolver(3, 4, (1, 1, 1))
    with pytest.raises(ExaHypeHipError):
        exa._lib.check(s.lib.exa_dg_plan_set_stage_a(s._plan, 7))
