"""The step as ONE kernel (include/exahype_hip.h exa_dg_corrector_predictor; 3-D, N = 6): Riemann solve + corrector of the previous step in
front of the predictor, against the two-kernel step and against the CPU oracle (relative 1e-10, the north-star's tolerance; ADER-DG is "parity
unpinned" against the reference, which holds no ADER-DG -- see tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from tests.util import euler_dg_state, rel_err

pytestmark = pytest.mark.gpu
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


@pytest.mark.parametrize("nc,n_it", [((2, 2, 2), -1), ((1, 1, 1), 2), ((3, 2, 1), 1), ((1, 3, 2), 3), ((9, 8, 8), -1)])
def test_one_kernel_steps_vs_oracle_and_two_kernel_steps(exa, orc, nc, n_it):
    from oracle.dg_operators import operators
    ops = operators(N)
    u = euler_dg_state(tuple(nc) + (N,) * 3, seed=77 + sum(nc))
    dx = [1.0 / nc[0], 0.8 / nc[1], 1.3 / nc[2]]
    dts = [0.02 * min(dx) / (2 * N - 1) * f for f in (1.0, 0.7, 1.2, 0.9)]      # a different dt every step: the corrector uses the previous one
    nit = N if n_it < 0 else n_it
    one = exa.AderDgSolver(3, N, nc, n_picard=n_it, dx=dx, one_kernel_step=True)
    two = exa.AderDgSolver(3, N, nc, n_picard=n_it, dx=dx, one_kernel_step=False)
    assert one._one_kernel and not two._one_kernel
    one.upload(u)
    two.upload(u)
    uo = u.reshape(-1).copy()
    steps = len(dts) if np.prod(nc) < 100 else 3
    for k in range(steps):
        one.step(dts[k])
        two.step(dts[k])
        assert one._pending_dt == dts[k]
        uo = orc.aderdg_step(uo, dts[k], dx, ops, 3, N, 5, orc.PDE_EULER, nit, nc)
        if k == 1:                                                     # reading u in the middle of a run applies the pending corrector
            assert rel_err(one.download().reshape(-1), uo) < 1e-10
            assert one._pending_dt is None
    a, b = one.download(), two.download()
    assert rel_err(a.reshape(-1), uo) < 1e-10
    assert rel_err(a, b) < 1e-12
    assert rel_err(one.trace.cpu().numpy(), two.trace.cpu().numpy()) < 1e-12


def test_one_kernel_step_is_refused_where_it_is_not_built(exa):
    with pytest.raises(ValueError):
        exa.AderDgSolver(3, 4, (2, 2, 2), one_kernel_step=True)
    with pytest.raises(ValueError):
        exa.AderDgSolver(3, N, (2, 2, 2), stage_a="lds", one_kernel_step=True)
    s = exa.AderDgSolver(3, N, (2, 2, 2))                              # off unless asked for
    assert not s._one_kernel
