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


def test_limited_step_refuses_cells_of_unequal_size():
    """The FV patch update takes ONE volume size (solvers.SubcellLimiter passes dx / N_s): a grid with dx[0] != dx[1] would get a silently wrong
    FV update -- the oracle's restatement of the glue refuses it, and so does the product (GPU test below)."""
    N, nc = 3, (2, 2)
    u = euler_dg_state(nc + (N, N), seed=3)
    mask = np.zeros(nc, bool)
    mask[0, 0] = True
    with pytest.raises(ValueError, match="one volume size"):
        limited_step(u, mask, 1e-3, [0.5, 0.4], operators(N), A.Euler(), _fv(2, 5, oracle.PDE_EULER))
    limited_step(u, mask, 1e-3, [0.5, 0.5], operators(N), A.Euler(), _fv(2, 5, oracle.PDE_EULER))


@pytest.mark.gpu
def test_subcell_limiter_refuses_cells_of_unequal_size():
    from exahype_amd import solvers as exa
    import torch
    mask = torch.zeros((2, 2), dtype=torch.bool, device="cuda")
    lim = exa.SubcellLimiter(exa.AderDgSolver(2, 3, (2, 2), dx=[0.5, 0.4]))      # (projection / reconstruction alone do not depend on the cell size)
    with pytest.raises(ValueError, match="one volume size"):
        lim.step(1e-3, mask)
    exa.SubcellLimiter(exa.AderDgSolver(2, 3, (2, 2), dx=[0.5, 0.5])).step(1e-3, mask)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,N,nc", [(2, 4, (4, 3)), (3, 3, (2, 2, 3)), (2, 2, (3, 3)),
                                       (3, 8, (2, 1, 2)),        # cfg 4's order: level-streamed stage A + 17^3 slab FV update + reconstruction
                                       (3, 7, (1, 2, 2))])
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
@pytest.mark.parametrize("dim,N,nc", [(3, 8, (2, 1, 2)), (2, 4, (3, 2)), (3, 3, (2, 2, 2))])
def test_all_variables_kernels_equal_the_per_variable_kernels(dim, N, nc, monkeypatch):
    """Five-variable systems take the all-variables projection / reconstruction kernels (AoS runs, one pass over the patch);
    EXA_LIM_PER_VARIABLE=1 sends them through the per-variable kernels every other variable count uses.  Same summation
    order: the limited step must come out bit for bit the same."""
    from exahype_amd import solvers as exa
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=77)
    dx = [1.0 / nc[0]] * dim
    dt = 0.02 * dx[0] / (2 * N - 1)
    mask = np.random.default_rng(3).random(nc) < 0.5
    mask.flat[0] = True
    res = []
    for per_variable in ("0", "1"):
        monkeypatch.setenv("EXA_LIM_PER_VARIABLE", per_variable)
        s = exa.AderDgSolver(dim, N, nc, dx=dx)
        lim = exa.SubcellLimiter(s)
        s.upload(u)
        for _ in range(2):
            lim.step(dt, mask)
        res.append(s.download())
    assert np.array_equal(res[0], res[1])


@pytest.mark.gpu
def test_limiter_capacity_and_device_resident_mask():
    """The troubled count never reaches the host inside step(): the cell list is compacted on the device into `capacity`
    slots.  A CUDA mask gives the same result as a numpy mask; more troubled cells than capacity is reported by check()."""
    import torch
    from exahype_amd import solvers as exa
    dim, N, nc = 2, 3, (4, 4)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=3)
    dx = [0.25] * dim
    dt = 0.02 * dx[0] / (2 * N - 1)
    mask = np.zeros(nc, dtype=bool)
    mask[0, 0] = mask[2, 1] = mask[3, 3] = True
    res = []
    for cap, m in ((None, mask), (3, torch.as_tensor(mask).cuda()), (5, mask)):
        s = exa.AderDgSolver(dim, N, nc, dx=dx)
        lim = exa.SubcellLimiter(s, capacity=cap)
        s.upload(u)
        n = lim.step(dt, m)
        assert int(n) == 3
        lim.check(wait=True)
        res.append(s.download())
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
    s = exa.AderDgSolver(dim, N, nc, dx=dx)
    lim = exa.SubcellLimiter(s, capacity=2)
    s.upload(u)
    lim.step(dt, mask)
    with pytest.raises(RuntimeError, match="capacity"):
        lim.check(wait=True)
    # the two cells that fit were limited, the third kept the DG result
    s2 = exa.AderDgSolver(dim, N, nc, dx=dx); s2.upload(u); s2.step(dt)
    a, b = s.download(), s2.download()
    assert np.array_equal(a[3, 3], b[3, 3]) and np.array_equal(a[0, 0], res[0][0, 0]) and np.array_equal(a[2, 1], res[0][2, 1])
    # the flag is sticky: a later step within the capacity does not hide the overflow, and check() reports it once
    s3 = exa.AderDgSolver(dim, N, nc, dx=dx)
    lim3 = exa.SubcellLimiter(s3, capacity=2)
    s3.upload(u)
    lim3.step(dt, mask)
    fewer = mask.copy()
    fewer[3, 3] = False
    try:                                                  # step() itself reports an overflow of a step the GPU has already completed
        lim3.step(dt, fewer)
        with pytest.raises(RuntimeError, match="capacity"):
            lim3.check(wait=True)
    except RuntimeError as e:
        assert "capacity" in str(e)
    lim3.check(wait=True)
    # ... and nobody has to ask: once the GPU is past an overflowing step, the next step() raises; reading the result back waits and raises
    s4 = exa.AderDgSolver(dim, N, nc, dx=dx)
    lim4 = exa.SubcellLimiter(s4, capacity=2)
    s4.upload(u)
    lim4.step(dt, mask)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="capacity"):
        lim4.step(dt, fewer)
    lim4.step(dt, mask)
    with pytest.raises(RuntimeError, match="capacity"):
        lim4.download()
    assert lim4.download().shape == u.shape
    # default capacity: a bounded share of the block, not the block
    big = exa.AderDgSolver(2, 3, (40, 40))
    assert exa.SubcellLimiter(big).capacity == 160


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


@pytest.mark.gpu
def test_troubled_cell_indicator():
    """Positivity + relaxed DMP: a smooth state is clean, a density jump / a negative pressure / an oscillation is flagged."""
    import torch
    from exahype_amd import solvers as exa
    dim, N, nc = 2, 4, (8, 6)
    ops = operators(N)
    xs = A.node_coords(nc, N, ops)
    u = np.zeros(nc + (N, N, 5))
    u[..., 0] = 1.0 + 0.1 * np.sin(2 * np.pi * xs[0]) * np.cos(2 * np.pi * xs[1])
    u[..., 1] = 0.3 * u[..., 0]
    u[..., 4] = 2.5 + 0.5 * 0.09 * u[..., 0]
    s = exa.AderDgSolver(dim, N, nc)
    lim = exa.SubcellLimiter(s)
    s.upload(u)
    assert not bool(lim.detect().any())
    v = u.copy()
    v[3, 2, 1, 2, 0] *= 1.8                                     # one node overshoots: that cell (its raised mean may
    s.upload(v)                                                 # narrow the admissible range of a face neighbour too)
    m = lim.detect().cpu().numpy()
    assert m[3, 2] and all(abs(i - 3) + abs(j - 2) <= 1 for i, j in np.argwhere(m))
    v = u.copy()
    v[5, 1, ..., 4] = 0.01                                      # negative pressure in one cell
    s.upload(v)
    m = lim.detect().cpu().numpy()
    assert m[5, 1] and m.sum() == 1
    v = u.copy()
    v[:4, ..., 0] = 1.0
    v[4:, ..., 0] = 0.125                                       # Sod-like jump between cells 3|4 (and the periodic wrap 7|0): smooth inside cells
    s.upload(v)
    m = lim.detect(dmp_tol=0.5).cpu().numpy()
    assert not m.any()                                          # piecewise constant data respects the DMP of the means
    v[4, :, 0, :, 0] = 0.6                                      # a Gibbs-like intermediate layer inside the cell at the jump ... is within range
    v[4, :, 1, :, 0] = -0.05                                    # ... an undershoot is not
    s.upload(v)
    m = lim.detect().cpu().numpy()
    assert m[4].all() and m.sum() == nc[1]
