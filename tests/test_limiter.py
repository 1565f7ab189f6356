"""FV subcell limiter glue (BASELINE configs[4]).  "Parity unpinned" against the reference (it has no
limiter): the oracle is oracle/limiter_numpy.py, pinned by the identities below; the FV update inside is the corrected
Rusanov patch update of oracle/exa_oracle.c."""
import numpy as np
import pytest

import oracle
from oracle import aderdg_numpy as A
from oracle.dg_operators import operators
from oracle.limiter_numpy import apply_all_axes, limited_step, projection_matrix, reconstruction_matrix
from tests.util import euler_dg_state, rel_err


@pytest.mark.parametrize("N", [2, 3, 4, 6, 8])
def test_projection_reconstruction_identities(N):
    o = operators(N)
    Ns = 2 * N - 1
    P = projection_matrix(o["xi"], Ns)
    R = reconstruction_matrix(P, o["w"])
    assert np.max(np.abs(R @ P - np.eye(N))) < 1e-12            # exact on degree <= p data
    assert np.max(np.abs(P.mean(axis=0) - o["w"])) < 1e-14       # projection preserves the cell mean
    assert np.max(np.abs(P.sum(axis=1) - 1)) < 1e-13             # constants stay constants
    v = np.random.default_rng(N).random(Ns)
    assert abs(o["w"] @ (R @ v) - v.mean()) < 1e-13              # reconstruction preserves the mean of ANY data


def _fv(dim, nv, pde):
    def update(patch, dt, h):
        Ns = patch.shape[0] - 2
        return oracle.fv_corrected(patch[None], dt, h, dim, Ns, 1, nv, 0, 1, pde)[0]
    return update


@pytest.mark.gpu
@pytest.mark.parametrize("dim,N,nc", [(2, 4, (4, 3)), (3, 3, (2, 2, 3)), (2, 2, (3, 3))])
def test_limited_step_vs_oracle(dim, N, nc):
    from exahype_amd import solvers as exa
    ops = operators(N)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=31 + N)
    dx = [1.0 / nc[0]] * dim                                    # uniform cells (the FV patch has one h)
    dt = 0.02 * dx[0] / (2 * N - 1)
    rng = np.random.default_rng(5)
    mask = rng.random(nc) < 0.35
    mask.flat[0] = True
    s = exa.AderDgSolver(dim, N, nc, dx=dx)
    lim = exa.SubcellLimiter(s)
    P, R = lim.operators()
    Po = projection_matrix(ops["xi"], 2 * N - 1)
    assert np.max(np.abs(P - Po)) < 1e-13 and np.max(np.abs(R - reconstruction_matrix(Po, ops["w"]))) < 1e-12
    s.upload(u)
    ref = u.copy()
    for _ in range(2):
        n = lim.step(dt, mask)
        assert n == int(mask.sum())
        ref = limited_step(ref, mask, dt, dx, ops, A.Euler(), _fv(dim, 5, oracle.PDE_EULER))
    assert rel_err(s.download(), ref) < 1e-10
    # untroubled cells are exactly the DG result, troubled cells differ from it
    s2 = exa.AderDgSolver(dim, N, nc, dx=dx); s2.upload(u); s2.step(dt)
    s.upload(u); lim.step(dt, mask)
    a, b = s.download(), s2.download()
    assert np.array_equal(a[~mask], b[~mask]) and not np.allclose(a[mask], b[mask], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_projection_then_reconstruction_is_identity_p7_3d():
    """cfg 4's sizes: 3-D, N = 8 (p = 7), N_s = 15 -- patch 17^3; project + reconstruct without an FV update returns u."""
    import ctypes as C
    import torch
    from exahype_amd import solvers as exa
    N, nc = 8, (2, 1, 2)
    u = euler_dg_state(nc + (N,) * 3, seed=2)
    s = exa.AderDgSolver(3, N, nc)
    s.upload(u)
    lim = exa.SubcellLimiter(s)
    cells = torch.arange(4, dtype=torch.int64, device="cuda")
    patches = torch.zeros((4, lim.patch_doubles), dtype=torch.float64, device="cuda")
    exa.check(s.lib.exa_dg_project_patches(s._plan, C.c_void_p(s.u.data_ptr()), C.c_void_p(cells.data_ptr()), 4, C.c_void_p(patches.data_ptr()), None))
    s.u.zero_()
    exa.check(s.lib.exa_dg_reconstruct_patches(s._plan, C.c_void_p(patches.data_ptr()), C.c_void_p(cells.data_ptr()), 4, C.c_void_p(s.u.data_ptr()), None))
    torch.cuda.synchronize()
    assert rel_err(s.download(), u) < 1e-12
    # interior of the patch == oracle projection; low-x halo of cell 0 == last layer of its periodic x-neighbour (cell 2)
    P = projection_matrix(operators(N)["xi"], 15)
    proj = apply_all_axes(P, u, 3, 3)
    pt = patches.cpu().numpy().reshape(4, 17, 17, 17, 5)
    assert rel_err(pt[0, 1:-1, 1:-1, 1:-1], proj[0, 0, 0]) < 1e-12
    assert rel_err(pt[0, 0, 1:-1, 1:-1], proj[1, 0, 0][14]) < 1e-12
    assert rel_err(pt[0, 1:-1, 1:-1, 16], proj[0, 0, 1][:, :, 0]) < 1e-12
