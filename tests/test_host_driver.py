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


@pytest.mark.parametrize("dim,grid,P,H", [(2, (3, 2), 4, 1), (2, (1, 2), 3, 2), (3, (2, 2, 3), 3, 1)])
def test_fill_halos_dirichlet_matches_global_array(dim, grid, P, H):
    """SURVEY.md 8(f)-3, non-periodic: patches cut from a global array that is padded with fixed boundary values."""
    from exahype_amd.solvers import fill_halos_dirichlet
    rng = np.random.default_rng(1)
    V = 3
    G = rng.random(tuple(g * P for g in grid) + (V,))
    states = {(a, s): rng.random(V) + 10 * (2 * a + s + 1) for a in range(dim) for s in range(2)}
    # global array with H boundary layers per side; axes padded in order, so edges / corners hold the last axis' state
    Gp = G
    for a in range(dim):
        lo_shape = list(Gp.shape); lo_shape[a] = H
        lo = np.broadcast_to(states[(a, 0)], lo_shape); hi = np.broadcast_to(states[(a, 1)], lo_shape)
        Gp = np.concatenate([lo, Gp, hi], axis=a)
    S = P + 2 * H
    want = np.zeros(tuple(grid) + (S,) * dim + (V,))
    for idx in np.ndindex(*grid):
        want[idx] = Gp[tuple(slice(idx[a] * P, idx[a] * P + S) for a in range(dim))]
    mask = np.zeros(want.shape[dim:2 * dim], dtype=bool)
    mask[(slice(H, H + P),) * dim] = True
    # what the (2 dim + 1)-point stencil reads: interiors and face halos (one axis out of the interior range at most)
    coords = np.indices((S,) * dim)
    outside = sum(((coords[a] < H) | (coords[a] >= H + P)).astype(int) for a in range(dim))
    read = outside <= 1
    for lib in ("numpy", "torch"):
        Q = want.copy()
        Q[(slice(None),) * dim + (~mask,)] = -7.0
        if lib == "torch":
            import torch
            Qt = torch.as_tensor(Q)
            fill_halos_dirichlet(Qt, grid, dim, P, H, states)
            Q = Qt.numpy()
        else:
            fill_halos_dirichlet(Q, grid, dim, P, H, states)
        assert np.array_equal(Q[(slice(None),) * dim + (read,)], want[(slice(None),) * dim + (read,)])
    # one fixed state for the whole boundary
    Q = want.copy()
    fill_halos_dirichlet(Q, grid, dim, P, H, np.full(V, 2.5))
    assert np.all(Q[(0,) * dim][(slice(0, H),) + (slice(H, H + P),) * (dim - 1)] == 2.5)
    assert np.array_equal(Q[(slice(None),) * dim + (mask,)], want[(slice(None),) * dim + (mask,)])


@pytest.mark.gpu
def test_fv_patch_grid_dirichlet_keeps_a_constant_state():
    """A uniform state with the same state prescribed on the boundary is a fixed point of the Dirichlet-driven loop."""
    from exahype_amd import solvers as exa
    grid, P, H = (2, 3), 4, 1
    state = np.array([1.2, 0.3, -0.2, 0.1, 2.9])
    fv = exa.FVPatchGrid(2, grid, P, H, 5, 0, exa.PDE_EULER, exa.FV_RUSANOV, boundary=state)
    fv.set_interior(np.broadcast_to(state, grid + (P, P, 5)).copy())
    for _ in range(3):
        fv.step(1e-3)
    assert np.max(np.abs(fv.interior() - state)) < 1e-14


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


@pytest.mark.gpu
@pytest.mark.parametrize("dim,N,nc", [(2, 3, (3, 4)), (2, 4, (5, 2)), (3, 4, (2, 3, 2)), (3, 6, (3, 2, 2)), (3, 8, (2, 1, 2)), (3, 5, (1, 2, 2))])
def test_stage_b_carries_the_next_cfl_scan(dim, N, nc):
    """exa_dg_riemann_corrector_cfl: the launch that writes the corrected u also leaves its largest eigenvalue -- bit-equal to the separate scan
    (exa_dg_max_eigenvalue) of the same u, and u itself bit-equal to the plain launch's; segmented and dense stage-B kernels, boxes with fewer cells than
    a workgroup holds; a NaN in the state arrives as a NaN."""
    import torch
    from exahype_amd import solvers as exa
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=50 + N)
    a, b = exa.AderDgSolver(dim, N, nc, n_vars=5), exa.AderDgSolver(dim, N, nc, n_vars=5)
    dt = 0.02 * min(a.dx) / (2 * N - 1)
    lam = torch.full((1,), -1.0, dtype=torch.float64, device="cuda")
    for s in (a, b):
        s.upload(u)
        s.predictor_volume(dt)
    a.riemann_corrector(dt)
    b.riemann_corrector(dt, lam_out=lam)
    assert torch.equal(a.u, b.u)
    assert float(lam[0]) == float(a.max_eigenvalue()[0]) > 0.0
    b.u.reshape(-1)[7] = float("nan")                                       # (the corrected u of the next launch then holds a NaN)
    b.predictor_volume(dt)
    b.riemann_corrector(dt, lam_out=lam)
    assert np.isnan(float(lam[0]))
    assert a.can_fuse_cfl_scan()


def test_cfl_step_rule():
    """dt = min(scale / lambda_max, time left); lambda_max == 0.0 exactly: one step to the end; NaN / inf / negative: the run has diverged and says so
    (r4 advice: fmax in the device reductions dropped a NaN and the loop went on with a finite lambda)."""
    from exahype_amd.solvers import _cfl_step
    assert _cfl_step(2.0, 1.0, 10.0, "t") == 0.5 and _cfl_step(2.0, 1.0, 0.25, "t") == 0.25 and _cfl_step(0.0, 1.0, 0.75, "t") == 0.75
    for bad in (float("nan"), float("inf"), -1.0):
        with pytest.raises(FloatingPointError, match="diverged"):
            _cfl_step(bad, 1.0, 1.0, "t")


@pytest.mark.gpu
def test_run_continues_from_the_current_time_and_raises_on_a_diverged_state():
    """Both solvers' run(t_end) integrate UNTIL `time` reaches t_end (r4: the DG loop integrated a duration from a local t = 0), and a NaN in the state
    reaches the host as a NaN eigenvalue (device reductions keep it: exa_pde.hpp nan_max) and ends the run with an error instead of silently."""
    import torch
    from exahype_amd import solvers as exa
    N, nc = 3, (3, 3)
    u = euler_dg_state(nc + (N, N), seed=4)
    s = exa.AderDgSolver(2, N, nc, n_vars=5, time=0.5)
    s.upload(u)
    lam = float(s.max_eigenvalue()[0])
    dt = 0.3 * min(s.dx) / ((2 * N - 1) * 2 * lam)
    assert s.run(0.5 + 1.5 * dt, cfl=0.3) == 2 and abs(s.time - (0.5 + 1.5 * dt)) < 1e-14
    assert s.run(0.5 + 1.5 * dt, cfl=0.3) == 0                                  # already there
    s.u[0, 0, 0, 0, 0] = float("nan")
    with pytest.raises(FloatingPointError, match="diverged"):
        s.run(s.time + dt, cfl=0.3)
    for fused in (True, False):
        fv = exa.FVPatchGrid(2, (3, 2), 4, 1, 5, 0, exa.PDE_EULER, exa.FV_RUSANOV, fused=fused)
        fv.set_interior(_grid_state((3, 2), 4, 5, 3, 2))
        t0 = fv.time
        assert fv.run(t0 + 1e-3, cfl=0.3) >= 1 and abs(fv.time - (t0 + 1e-3)) < 1e-15
        U = fv.interior()
        U[1, 1, 2, 2, 0] = float("nan")
        fv.set_interior(U)
        with pytest.raises(FloatingPointError, match="diverged"):
            fv.run(fv.time + 1e-3, cfl=0.3)


def _grid_state(grid, P, V, seed, dim):
    """Euler-like admissible states [g.., P.., V]"""
    rng = np.random.default_rng(seed)
    Q = np.zeros(tuple(grid) + (P,) * dim + (V,))
    Q[..., 0] = 1.0 + 0.3 * rng.random(Q.shape[:-1])
    for a in range(1, 4):
        Q[..., a] = Q[..., 0] * (0.4 * rng.random(Q.shape[:-1]) - 0.2)
    Q[..., 4] = 2.6 + 0.5 * rng.random(Q.shape[:-1])
    if V > 5:
        Q[..., 5:] = rng.random(Q.shape[:-1] + (V - 5,))
    return Q


@pytest.mark.gpu
@pytest.mark.parametrize("dim,grid,P,H,n_aux,mode,dirichlet", [
    (2, (64, 1024), 4, 1, 5, "faithful", False),    # the reference's shape: staged, persistent grid (>= 2048 blocks of 16 patches)
    (2, (5, 3), 4, 1, 5, "faithful", True),         # same shape, few patches: staged, one pass; ragged last block, wraps inside a block
    (2, (3, 4), 4, 1, 0, "rusanov", True),          # no auxiliary variables: per-volume writes
    (2, (2, 3), 6, 2, 1, "rusanov", False),         # two halo layers, odd variable count
    (2, (2, 2), 40, 1, 0, "rusanov", False),        # 1 600 volumes per patch: no LDS copy, the stencil reads the neighbour patch directly
    (3, (2, 3, 2), 4, 1, 0, "rusanov", True),       # 3-D small patches, staged
    (3, (3, 2, 2), 15, 1, 0, "rusanov", False),     # cfg 4's limiter patch: plane-streaming kernel, cached scalars
    (3, (2, 2, 3), 15, 1, 0, "faithful", True),     # ... its faithful form, domain faces with prescribed states
    (3, (1, 2, 1), 12, 1, 2, "rusanov", False),     # a grid extent of 1: the patch is its own neighbour
    (2, (1, 1), 4, 1, 5, "rusanov", False),         # ONE patch: every halo is its own far side
    (2, (1, 7), 4, 1, 5, "rusanov", True),          # fewer patches than a block holds; extent 1 along the slow axis
    (3, (2, 1, 3), 5, 2, 1, "rusanov", True),       # 3-D, two halo layers, 125 volumes per patch (two patches per block)
    (2, (3, 2), 16, 1, 0, "rusanov", False),        # one patch per 256-thread workgroup, staged
    (2, (2, 2), 24, 1, 3, "faithful", True),        # one patch per 1024-thread workgroup, staged
    (3, (2, 2, 2), 6, 1, 10, "rusanov", False),     # staged, 3 240 remote halo units per block > the 3 072 the decoded path's registers cover (r4 advice)
    (3, (2, 2, 1), 6, 2, 3, "rusanov", True),       # ... 3 456 units (two halo layers, 8 variables)
])
def test_grid_step_equals_halo_fill_plus_patch_update(dim, grid, P, H, n_aux, mode, dirichlet):
    """exa_fv_grid_step_device (halo-less arrays; the states beyond a patch face taken from the neighbours inside the launch) is BIT-equal to
    the two-pass form it replaces -- halo fill of an array with halo, then the in-place `time_step` -- for every kernel variant the dispatch
    knows, periodic and with prescribed boundary states, over several steps (so the array swap is exercised)."""
    from exahype_amd import solvers as exa
    V = 5 + n_aux
    pde = exa.PDE_EULER_REF2D if (dim == 2 and mode == "faithful") else exa.PDE_EULER
    m = exa.FV_FAITHFUL if mode == "faithful" else exa.FV_RUSANOV
    rng = np.random.default_rng(7)
    bnd = None
    if dirichlet:
        bnd = {(a, s_): np.concatenate([[1.1 + 0.1 * a, 0.1, -0.05 * s_, 0.02, 2.8], rng.random(n_aux)]) for a in range(dim) for s_ in range(2)}
    U0 = _grid_state(grid, P, V, 11, dim)
    a = exa.FVPatchGrid(dim, grid, P, H, 5, n_aux, pde, m, boundary=bnd, fused=True)
    b = exa.FVPatchGrid(dim, grid, P, H, 5, n_aux, pde, m, boundary=bnd, fused=False)
    a.set_interior(U0); b.set_interior(U0)
    assert a.U.shape == tuple(grid) + (P,) * dim + (V,)                       # no halo bytes in HBM
    dt = 1e-3 if mode == "rusanov" else 1e-4
    for k in range(3):
        a.step(dt)
        b.step(dt)
        ia, ib = a.interior(), b.interior()
        assert np.isfinite(ib).all()
        assert np.array_equal(ia, ib), (k, float(np.nanmax(np.abs(ia - ib))))
    assert abs(a.time - b.time) == 0.0
    # the layout with halo on demand: interiors + the halo layers the two-pass form had filled for this step
    b.fill_halos()
    S = P + 2 * H
    co = np.indices((S,) * dim)
    read = sum(((co[x] < H) | (co[x] >= H + P)).astype(int) for x in range(dim)) <= 1        # (corners / edges: not part of the stencil)
    assert np.array_equal(a.with_halo().cpu().numpy()[(slice(None),) * dim + (read,)], b.Q.cpu().numpy()[(slice(None),) * dim + (read,)])
    with pytest.raises(AttributeError, match="halo-less"):
        a.Q
    # the CFL scan: left behind by the kernel that wrote the states == a scan pass over the array == the host maximum
    lam_fused = a.max_eigenvalue()
    a.invalidate()
    lam_scan = a.max_eigenvalue()
    q = a.interior().reshape(-1, V)
    want = max(float(np.max(exa.pde_eval(pde, d, q)[1])) for d in range(dim))
    if dirichlet:
        want = max(want, max(float(np.max(exa.pde_eval(pde, d, np.stack(list(bnd.values())))[1])) for d in range(dim)))
    assert lam_scan == want
    assert abs(lam_fused - want) <= 1e-12 * want          # (the plane-streaming kernel evaluates it with its fast reciprocal / square root)


@pytest.mark.gpu
def test_grid_step_arguments_are_checked():
    import ctypes as C
    import torch
    from exahype_amd import solvers as exa
    fv = exa.FVPatchGrid(2, (3, 2), 4, 1, 5, 0)
    lib, plan = fv.lib, fv.kernel._plan
    U = fv.U
    other = torch.zeros_like(U)
    grid = exa.larr([3, 2])
    call = lambda q, qn, g: lib.exa_fv_grid_step_device(plan, C.c_void_p(q.data_ptr()), C.c_void_p(qn.data_ptr()), g, None, None, 0.0, 1e-3, fv.h, None, None)
    assert call(U, U, grid) != 0 and b"array of their own" in lib.exa_last_error()
    assert call(U, other, exa.larr([3, 3])) != 0 and b"patches" in lib.exa_last_error()
    assert call(U, other, exa.larr([0, 6])) != 0
    assert call(U, other, grid) == 0
    # halo wider than the patch: the layers would reach past the face neighbour
    wide = exa.FVRusanovKernel(2, 2, 3, 5, 0, 6, exa.PDE_EULER, exa.FV_RUSANOV)
    assert lib.exa_fv_grid_step_device(wide._plan, C.c_void_p(U.data_ptr()), C.c_void_p(other.data_ptr()), grid, None, None, 0.0, 1e-3, 0.1, None, None) != 0
    assert b"halo_size" in lib.exa_last_error()


@pytest.mark.gpu
def test_grid_step_with_a_generated_term_set_and_cfl_loop():
    """The grid step through a user library (exa_user_fv_launch with the grid arguments) == the two-pass form; run() with the fused CFL scan takes
    the same steps as run() with a scan pass per step."""
    from exahype_amd import solvers as exa
    from tests.test_user_pde import swe, swe_state
    p = swe()
    grid, P = (5, 4), 6
    U0 = swe_state(grid + (P, P), 4)
    a = exa.FVPatchGrid(2, grid, P, 1, 3, 0, p.register(), exa.FV_RUSANOV, fused=True)
    b = exa.FVPatchGrid(2, grid, P, 1, 3, 0, p.register(), exa.FV_RUSANOV, fused=False)
    a.set_interior(U0); b.set_interior(U0)
    t_end = 0.02
    na, nb_ = a.run(t_end, cfl=0.3), b.run(t_end, cfl=0.3)
    assert na == nb_ and na >= 3 and abs(a.time - t_end) < 1e-15 and abs(b.time - t_end) < 1e-15
    # the fused scan evaluates the eigenvalue with the same (IEEE) members on the same states: the same dt sequence, the same result
    assert np.array_equal(a.interior(), b.interior())
