"""the steps either side of the kernel -- periodic halo fill of a patch grid and the
CFL time loops.  The halo fill is host logic (CPU test against a global periodic array); the loops run
on the GPU against the oracle."""
import numpy as np
import pytest

from tests.util import euler_dg_state, rel_err


def _scatter(global_arr, grid, P, H):
    """global [G0*P, G1*P, V] -> patches [g0, g1, S, S, V] with periodic halos taken from the global array."""
    dim = len(grid)
    S = P + 2 * H
    V = global_arr.shape[-1]
    out = np.zeros(tuple(grid) + (S,) * dim + (V,))
    for idx in np.ndindex(*grid):
        sl = []
        for a in range(dim):
            ix = (np.arange(-H, P + H) + idx[a] * P) % (grid[a] * P)
            sl.append(ix)
        out[idx] = global_arr[np.ix_(*sl)]
    return out


@pytest.mark.parametrize("dim,grid,P,H", [(2, (3, 2), 4, 1), (2, (2, 2), 3, 2), (3, (2, 3, 2), 3, 1)])
def test_fill_halos_periodic_matches_global_array(dim, grid, P, H):
    from exahype_amd.solvers import fill_halos_periodic
    rng = np.random.default_rng(0)
    G = rng.random(tuple(g * P for g in grid) + (3,))
    want = _scatter(G, grid, P, H)
    Q = want.copy()
    # wipe every halo, keep interiors
    mask = np.zeros(Q.shape[dim:2 * dim], dtype=bool)
    mask[(slice(H, H + P),) * dim] = True
    Q[(slice(None),) * dim + (~mask,)] = -7.0
    fill_halos_periodic(Q, grid, dim, P, H)
    assert np.array_equal(Q, want)
    import torch
    Qt = torch.as_tensor(want.copy())
    Qt[(slice(None),) * dim + (torch.as_tensor(~mask),)] = -7.0
    fill_halos_periodic(Qt, grid, dim, P, H)
    assert np.array_equal(Qt.numpy(), want)


@pytest.mark.gpu
def test_fv_patch_grid_equals_one_big_periodic_patch():
    """A 3x2 grid of 4x4 patches advanced with halo fills == the oracle's update of the one 12x8 periodic patch."""
    import oracle
    from exahype_amd import solvers as exa
    grid, P, H = (3, 3), 4, 1
    n = grid[0] * P
    G = euler_dg_state((n, n), seed=5, amp=0.3)
    fv = exa.FVPatchGrid(2, grid, P, H, 5, 0, exa.PDE_EULER, exa.FV_RUSANOV)
    interior = np.zeros(grid + (P, P, 5))
    for idx in np.ndindex(*grid):
        interior[idx] = G[idx[0] * P:(idx[0] + 1) * P, idx[1] * P:(idx[1] + 1) * P]
    fv.set_interior(interior)
    dt = 0.2 * fv.h / 2 / 2.5
    big = np.zeros((1, n + 2, n + 2, 5))
    cur = G.copy()
    for _ in range(3):
        fv.step(dt)
        big[0, 1:-1, 1:-1] = cur
        big[0, 0, 1:-1] = cur[-1]; big[0, -1, 1:-1] = cur[0]; big[0, 1:-1, 0] = cur[:, -1]; big[0, 1:-1, -1] = cur[:, 0]
        cur = oracle.fv_corrected(big, dt, fv.h, 2, n, 1, 5, 0, pde=oracle.PDE_EULER)[0, 1:-1, 1:-1]
    got = fv.interior()
    for idx in np.ndindex(*grid):
        assert rel_err(got[idx], cur[idx[0] * P:(idx[0] + 1) * P, idx[1] * P:(idx[1] + 1) * P]) < 1e-12
    steps = fv.run(fv.time + 3 * dt, cfl=0.3)
    assert steps >= 1 and np.isfinite(fv.interior()).all()


@pytest.mark.gpu
def test_dg_cfl_time_loop():
    import oracle
    from oracle.dg_operators import operators
    from exahype_amd import solvers as exa
    N, nc = 3, (3, 3)
    u = euler_dg_state(nc + (N, N), seed=4)
    s = exa.AderDgSolver(2, N, nc, n_vars=5)
    s.upload(u)
    lam = float(s.max_eigenvalue()[0])
    t_end = 2.5 * 0.3 * min(s.dx) / ((2 * N - 1) * 2 * lam)
    steps = s.run(t_end, cfl=0.3)
    assert steps == 3
    # replay on the oracle with the same adaptive rule
    ref, t = u.reshape(-1).copy(), 0.0
    L = oracle.lib()
    for _ in range(steps):
        q = ref.reshape(-1, 5)
        lam_o = max(L.orc_pde_maxeig(oracle.PDE_EULER, np.ascontiguousarray(q[i]), d) for i in range(len(q)) for d in range(2))
        dt = min(0.3 * min(s.dx) / ((2 * N - 1) * 2 * lam_o), t_end - t)
        ref = oracle.aderdg_step(ref, dt, s.dx, operators(N), 2, N, 5, oracle.PDE_EULER, N, nc)
        t += dt
    assert rel_err(s.download().reshape(-1), ref) < 1e-10
