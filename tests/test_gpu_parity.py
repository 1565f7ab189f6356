"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs, and against the committed golden vectors of the reference.

Tolerances: FV faithful mode is compared BIT-EXACT (the FV unit is built with
-ffp-contract=off, like the g++ build of the reference); everything else is fp64
with FMA contraction on the device: relative 1e-10 as the north-star states
(observed ~1e-14).  ADER-DG results are "parity unpinned" against the reference
(it has no ADER-DG): the oracle for them is oracle/exa_oracle.c,
itself pinned by the KATs in tests/test_aderdg_kat.py.
"""
import json
import os

import numpy as np
import pytest

from tests.util import euler_dg_state, euler_patches, euler_ref2d_patches, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def exa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import exahype_amd
    from exahype_amd import solvers
    assert exahype_amd._lib.device_count() >= 1
    return solvers


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.lib()
    return oracle


# ---- point-wise PDE terms (Functions.cpp) -------------------------------------------------------
def test_pde_terms_vs_reference_golden(exa, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "euler_terms_ref2d.json")))
    Q = np.array(g["Q"]); fl = np.array(g["flux"]); ev = np.array(g["maxeig"])
    for d in range(2):
        F, lam = exa.pde_eval(exa.PDE_EULER_REF2D, d, Q)
        assert np.array_equal(lam, ev[:, d]) or rel_err(lam, ev[:, d]) < 1e-15
        assert np.array_equal(F[:, :4], fl[:, d]) or rel_err(F[:, :4], fl[:, d]) < 1e-15


def test_pde_terms_euler3_vs_oracle(exa, orc):
    Q = euler_dg_state((257,), 5).reshape(-1, 5)
    for d in range(3):
        F, lam = exa.pde_eval(exa.PDE_EULER, d, Q)
        Fo = np.zeros_like(Q); lo = np.zeros(len(Q))
        for i in range(len(Q)):
            f = np.zeros(5); orc.lib().orc_pde_flux(orc.PDE_EULER, 5, np.ascontiguousarray(Q[i]), d, f); Fo[i] = f
            lo[i] = orc.lib().orc_pde_maxeig(orc.PDE_EULER, np.ascontiguousarray(Q[i]), d)
        assert rel_err(F, Fo) < 1e-14 and rel_err(lam, lo) < 1e-14


# ---- FV Rusanov patch update == time_step (test.cpp) ----------------------------------------------
def test_fv_faithful_sin_input_vs_reference_golden(exa, golden_dir):
    """The reference's own protocol: Q = sin(3.141 i/N), dt = 1 (correctness_test.cpp:102-106,195)."""
    g = json.load(open(os.path.join(golden_dir, "fv_ref2d_sin.json")))
    Q = np.sin(3.141 * np.arange(360) / 360)
    Q0 = Q.copy()
    k = exa.FVRusanovKernel(2, 4, 1, 5, 5, 1, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    k.time_step(Q, 1.0)
    idx = np.array(g["valid_modified_idx"]); val = np.array(g["valid_modified_val"])
    assert np.array_equal(Q[idx], val), np.abs(Q[idx] - val).max()          # bit-exact on the defined outputs
    pidx = np.array(g["passthrough_idx"])
    assert np.array_equal(Q[pidx], Q0[pidx])                                  # halo + aux vars untouched


def test_fv_faithful_random_vs_reference_golden(exa, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "fv_ref2d_random.json")))
    Q = np.array(g["Q_in"]); want = np.array(g["Q_out_valid"])
    k = exa.FVRusanovKernel(2, 4, 1, 5, 5, Q.shape[0], exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    out = np.ascontiguousarray(Q.copy())
    k.time_step(out, g["dt"])
    assert np.array_equal(out[:, 2:4, 2:4, 0:4], want)


@pytest.mark.parametrize("n_patches,P,H,n_real,n_aux", [(1, 4, 1, 5, 5), (37, 4, 1, 5, 5), (5, 7, 2, 4, 0), (3, 20, 1, 5, 2), (2, 40, 1, 4, 1)])
def test_fv_faithful_vs_oracle_2d(exa, orc, n_patches, P, H, n_real, n_aux):
    S, V = P + 2 * H, n_real + n_aux
    Q = euler_ref2d_patches(n_patches, S, V, seed=P * 100 + n_patches)
    want = orc.fv_faithful(Q, 0.3, 2, P, H, n_real, n_aux, n_patches, orc.PDE_EULER_REF2D)
    k = exa.FVRusanovKernel(2, P, H, n_real, n_aux, n_patches, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    got = np.ascontiguousarray(Q.copy()); k.time_step(got, 0.3)
    assert np.array_equal(got, want), rel_err(got, want)


@pytest.mark.parametrize("dim,P,H,n_aux,n_patches", [(2, 8, 1, 0, 9), (3, 4, 1, 0, 6), (3, 15, 1, 0, 2), (3, 6, 2, 3, 3),
                                                      (3, 11, 2, 1, 2), (3, 16, 1, 0, 1), (3, 13, 1, 2, 3)])   # plane-streaming kernel: halo 2, aux vars, odd/even planes
def test_fv_faithful_and_corrected_vs_oracle_euler(exa, orc, dim, P, H, n_aux, n_patches):
    S, V = P + 2 * H, 5 + n_aux
    Q = euler_patches(n_patches, dim, S, V, seed=dim * 10 + P)
    k = exa.FVRusanovKernel(dim, P, H, 5, n_aux, n_patches, exa.PDE_EULER, exa.FV_FAITHFUL)
    got = np.ascontiguousarray(Q.copy()); k.time_step(got, 0.05)
    want = orc.fv_faithful(Q, 0.05, dim, P, H, 5, n_aux, n_patches, orc.PDE_EULER)
    assert np.array_equal(got, want), rel_err(got, want)
    k = exa.FVRusanovKernel(dim, P, H, 5, n_aux, n_patches, exa.PDE_EULER, exa.FV_RUSANOV)
    got = np.ascontiguousarray(Q.copy()); k.time_step(got, 0.01, 0.1)
    want = orc.fv_corrected(Q, 0.01, 0.1, dim, P, H, 5, n_aux, n_patches, orc.PDE_EULER)
    assert rel_err(got, want) < TOL


def test_fv_device_resident_and_empty(exa, orc):
    import torch
    Q = euler_ref2d_patches(4096, 6, 10, seed=3)
    k = exa.FVRusanovKernel(2, 4, 1, 5, 5, 4096, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    Qd = torch.as_tensor(Q).cuda()
    k.time_step(Qd, 0.2)
    want = orc.fv_faithful(Q, 0.2, 2, 4, 1, 5, 5, 4096, orc.PDE_EULER_REF2D)
    assert np.array_equal(Qd.cpu().numpy(), want)
    k0 = exa.FVRusanovKernel(2, 4, 1, 5, 5, 0, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    k0.time_step(np.zeros((0, 6, 6, 10)), 0.1)


@pytest.mark.parametrize("dim,P,H,n_aux,mode", [(2, 4, 1, 5, 0), (3, 15, 1, 0, 1), (3, 5, 1, 1, 1), (2, 20, 1, 0, 1)])
def test_fv_masked_patches_are_skipped(exa, orc, dim, P, H, n_aux, mode):
    """exa_fv_time_step_device_masked: patches whose slot entry is negative stay bit for bit as they were (every kernel
    variant: staged small patches, register-resident, plane-streaming); the others equal the unmasked update."""
    import torch
    n = 7
    S, V = P + 2 * H, 5 + n_aux
    Q = euler_patches(n, dim, S, V, seed=4 + P)
    pde = exa.PDE_EULER
    k = exa.FVRusanovKernel(dim, P, H, 5, n_aux, n, pde, mode)
    full = torch.as_tensor(Q).cuda()
    k.time_step(full, 0.01, 0.1)
    slot = torch.tensor([0, -1, 5, -1, -1, 2, 9], dtype=torch.int64, device="cuda")
    part = torch.as_tensor(Q).cuda()
    k.time_step(part, 0.01, 0.1, slot=slot)
    a, b = part.cpu().numpy(), full.cpu().numpy()
    on = slot.cpu().numpy() >= 0
    assert np.array_equal(a[on], b[on]) and np.array_equal(a[~on], Q[~on]) and not np.array_equal(a[on], Q[on])


def test_fv_rejects_bad_config(exa):
    from exahype_amd._lib import ExaHypeHipError
    for kw in (dict(dim=1, patch_size=4, halo_size=1), dict(dim=2, patch_size=0, halo_size=1), dict(dim=2, patch_size=4, halo_size=0)):
        with pytest.raises(ExaHypeHipError):
            exa.FVRusanovKernel(n_real=5, n_aux=0, **kw)


# ---- ADER-DG -----------------------------------------------------------------------------------
DG_CASES = [(2, 4, (5, 3)), (2, 2, (4, 4)), (2, 8, (2, 3)), (2, 5, (3, 2)), (2, 3, (2, 4)), (2, 7, (2, 2)), (3, 3, (3, 2, 2)), (3, 4, (2, 2, 3)), (3, 5, (2, 1, 3)), (3, 6, (2, 2, 2)),
            (3, 7, (2, 1, 2)), (3, 8, (1, 2, 2)),       # N = 7, 8 in 3-D (cfg 4's p = 7): level-streamed stage A (exa_dg_stream.hpp)
            (3, 2, (2, 2, 2)), (2, 6, (3, 2)), (3, 6, (1, 1, 1)), (2, 4, (1, 1))]     # lowest order; a single periodic cell (its own neighbour everywhere)


def _ops(N):
    from oracle.dg_operators import operators
    return operators(N)


@pytest.mark.parametrize("dim,N,nc", DG_CASES)
def test_dg_operators_match_oracle(exa, dim, N, nc):
    s = exa.AderDgSolver(dim, N, nc)
    mine, ref = s.operators(), _ops(N)
    for k in ("xi", "w", "D", "Kxi", "phiL", "phiR", "iK1"):
        assert np.max(np.abs(mine[k] - ref[k])) < 2e-13, k


@pytest.mark.parametrize("dim,N,nc", DG_CASES)
@pytest.mark.parametrize("n_it", [-1, 0])
def test_dg_stage_a_and_step_vs_oracle(exa, orc, dim, N, nc, n_it):
    ops = _ops(N)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=dim * 100 + N)
    dx = [1.0 / c for c in nc]
    dt = 0.02 * min(dx) / (2 * N - 1)
    nit = N if n_it < 0 else 0
    s = exa.AderDgSolver(dim, N, nc, n_picard=n_it, dx=dx)
    s.upload(u)
    s.predictor_volume(dt)
    us_o, tr_o = orc.aderdg_stage_a(u.reshape(-1), dt, dx, ops, dim, N, 5, orc.PDE_EULER, nit)
    us = s.download()
    tr = s.trace.cpu().numpy().reshape(tr_o.shape)
    assert rel_err(us.reshape(-1), us_o) < TOL
    assert rel_err(tr, tr_o) < TOL
    s.riemann_corrector(dt)
    un_o = orc.aderdg_stage_b(us_o, tr_o, dt, dx, ops, dim, N, 5, orc.PDE_EULER, nc)
    assert rel_err(s.download().reshape(-1), un_o) < TOL
    # several steps, end to end
    s.upload(u)
    uo = u.reshape(-1).copy()
    for _ in range(3):
        s.step(dt)
        uo = orc.aderdg_step(uo, dt, dx, ops, dim, N, 5, orc.PDE_EULER, nit, nc)
    assert rel_err(s.download().reshape(-1), uo) < TOL


@pytest.mark.parametrize("N,nc,box", [(3, (8, 16, 8), None), (6, (8, 8, 8), None), (4, (16, 8, 24), ((8, 0, 8), (16, 8, 24))), (8, (8, 8, 8), None)])
def test_dg_stage_b_tile_order_vs_oracle(exa, orc, N, nc, box):
    """stage B enumerates the cells of a box in 8^3 tiles where every extent is a multiple of 8 (dg_inst.hip launch_b): dense (Nf = 9, 36) and
    segmented-shuffle (Nf = 16, 64) kernels, a whole block and a sub-box whose extents are multiples of 8 beside one that is not"""
    dim = 3
    ops = _ops(N)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=900 + N)
    dx = [1.0 / c for c in nc]
    dt = 0.02 * min(dx) / (2 * N - 1)
    s = exa.AderDgSolver(dim, N, nc, dx=dx)
    s.upload(u)
    s.predictor_volume(dt)
    us_o, tr_o = orc.aderdg_stage_a(u.reshape(-1), dt, dx, ops, dim, N, 5, orc.PDE_EULER, N)
    if box is None:
        s.riemann_corrector(dt)
    else:                                                             # the tiled sub-box first, the rest of the block lexicographically
        lo, hi = box
        s.riemann_corrector(dt, lo, hi)
        s.riemann_corrector(dt, (0, 0, 0), (lo[0], nc[1], nc[2]))
        s.riemann_corrector(dt, (lo[0], 0, 0), (hi[0], nc[1], lo[2]))
    un_o = orc.aderdg_stage_b(us_o, tr_o, dt, dx, ops, dim, N, 5, orc.PDE_EULER, nc)
    assert rel_err(s.download().reshape(-1), un_o) < TOL


def test_dg_box_launches_cover_block(exa, orc):
    """stage A / stage B over disjoint boxes == one launch over the block."""
    dim, N, nc = 3, 4, (4, 3, 5)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=77)
    dt = 1e-3
    a = exa.AderDgSolver(dim, N, nc); a.upload(u); a.step(dt)
    b = exa.AderDgSolver(dim, N, nc); b.upload(u)
    part = exa.CartesianPartition(8, 0, 3)
    shell, interior = part.shell_and_interior(nc)
    for lo, hi in shell + [interior]:
        b.predictor_volume(dt, lo, hi)
    for lo, hi in shell + [interior]:
        b.riemann_corrector(dt, lo, hi)
    # (a box launch enumerates the cells in another order than the block launch; the arithmetic per cell is the same)
    assert rel_err(a.download(), b.download()) < 1e-13


@pytest.mark.parametrize("N,nc", [(6, (12, 10, 9)), (8, (7, 7, 6)), (4, (13, 11, 9)), (5, (9, 8, 8))])
def test_dg_persistent_grid_multipass_vs_oracle(exa, orc, N, nc):
    """More cell blocks than resident workgroups, and not a multiple of them: the persistent grid of stage A walks
    over several blocks per workgroup (`blk += gridDim.x`: cell-id ping-pong, prefetch of the next block's u, LDS
    image reused) -- the regime the 128^3 benchmark runs in.  1 080 cells at N = 6 and 294 at N = 8 against 256
    resident workgroups; 322 blocks of 4 cells at N = 4, 288 blocks of 2 cells at N = 5 (CPB > 1)."""
    ops = _ops(N)
    u = euler_dg_state(tuple(nc) + (N,) * 3, seed=500 + N)
    dx = [1.0 / c for c in nc]
    dt = 0.02 * min(dx) / (2 * N - 1)
    s = exa.AderDgSolver(3, N, nc, dx=dx)
    s.upload(u)
    s.predictor_volume(dt)
    us_o, tr_o = orc.aderdg_stage_a(u.reshape(-1), dt, dx, ops, 3, N, 5, orc.PDE_EULER, N)
    assert rel_err(s.download().reshape(-1), us_o) < TOL                       # a stale / missing u* write-back shows here
    assert rel_err(s.trace.cpu().numpy().reshape(tr_o.shape), tr_o) < TOL
    s.upload(u)
    uo = u.reshape(-1).copy()
    for _ in range(2):
        s.step(dt)
        uo = orc.aderdg_step(uo, dt, dx, ops, 3, N, 5, orc.PDE_EULER, N, nc)
    assert rel_err(s.download().reshape(-1), uo) < TOL


def test_dg_box_launches_multipass_n6(exa, orc):
    """cfg 3's order on shell / interior boxes of a block with more cells than resident workgroups: stage A (N = 6,
    persistent grid) over the boxes of the 2x2x2 partition + stage B box by box == the oracle's step of the block."""
    N, nc = 6, (9, 8, 7)
    ops = _ops(N)
    u = euler_dg_state(tuple(nc) + (N,) * 3, seed=606)
    dx = [1.0 / c for c in nc]
    dt = 0.02 * min(dx) / (2 * N - 1)
    b = exa.AderDgSolver(3, N, nc, dx=dx); b.upload(u)
    shell, interior = exa.CartesianPartition(8, 0, 3).shell_and_interior(nc)
    for lo, hi in shell + [interior]:
        b.predictor_volume(dt, lo, hi)
    for lo, hi in shell + [interior]:
        b.riemann_corrector(dt, lo, hi)
    want = orc.aderdg_step(u.reshape(-1), dt, dx, ops, 3, N, 5, orc.PDE_EULER, N, nc)
    assert rel_err(b.download().reshape(-1), want) < TOL


def test_dg_advection_polynomial_exactness(exa):
    """KAT A.5-4 on the GPU: a global polynomial of degree <= p is advected exactly."""
    dim, N, nc = 2, 4, (6, 6)
    ops = _ops(N)
    from oracle import aderdg_numpy as A
    xs = A.node_coords(nc, N, ops)
    a = np.array([1.0, 0.5])
    poly = lambda x, y: 1 + x - 2 * y + x * y + x ** 3 - y ** 2 * x
    u = poly(xs[0], xs[1])[..., None] * np.ones((1,) * 4 + (1,))
    s = exa.AderDgSolver(dim, N, nc, pde=exa.PDE_ADVECTION, n_vars=1)
    dt = 0.02
    s.upload(np.broadcast_to(u, tuple(nc) + (N, N, 1)).copy())
    s.step(dt)
    got = s.download()[..., 0]
    want = poly(xs[0] - a[0] * dt, xs[1] - a[1] * dt) * np.ones_like(got)
    inner = (slice(1, -1), slice(1, -1))          # cells whose upwind neighbours are not across the wrap
    assert np.max(np.abs(got[inner] - want[inner])) < 1e-12


def test_dg_max_eigenvalue(exa, orc):
    dim, N, nc = 3, 3, (2, 2, 2)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=9)
    s = exa.AderDgSolver(dim, N, nc); s.upload(u)
    got = float(s.max_eigenvalue().cpu()[0])
    q = u.reshape(-1, 5)
    want = max(orc.lib().orc_pde_maxeig(orc.PDE_EULER, np.ascontiguousarray(q[i]), d) for i in range(len(q)) for d in range(3))
    assert abs(got - want) < 1e-13 * want


def test_dg_rejects_unsupported(exa):
    from exahype_amd._lib import ExaHypeHipError
    with pytest.raises(ExaHypeHipError):
        exa.AderDgSolver(3, 9, (2, 2, 2))          # N > 8 is not built
    with pytest.raises(ExaHypeHipError):
        exa.AderDgSolver(3, 4, (2, 2, 2), n_vars=4)


# ---- through the operator surface --------------------------------------------------------------------
def test_hip_printer_runs_builder_script(exa, orc):
    """A Rusanov patch-update script written against the operator surface (own names, 3-D, 5+2 variables, 3 patches) ->
    HIPPrinter -> fused HIP kernel; expected output: the oracle's faithful restatement (pinned bit for bit against the
    compiled reference, tests/test_oracle_pinning.py), compared BIT-EXACT."""
    from exahype_amd import KernelBuilder
    from exahype_amd.printers import HIPPrinter
    from tests.ref_examples import rusanov_patch_update
    dim, P, H, m, aux, n = 3, 6, 1, 5, 2, 3
    p = HIPPrinter(rusanov_patch_update(KernelBuilder, dim, P, H, m, aux, n))
    assert p.scheme == "fv-rusanov-faithful" and p.pde == 1
    Q = euler_patches(n, dim, P + 2 * H, m + aux, seed=99)
    want = orc.fv_faithful(Q, 0.07, dim, P, H, m, aux, n, orc.PDE_EULER)
    got = np.ascontiguousarray(Q.copy())
    p.run(got, 0.07)
    assert np.array_equal(got, want)
    # and the reference's own configuration (2-D, P=4, H=1, 5+5), same script: its golden vector
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fv_ref2d_sin.json")))
    p2 = HIPPrinter(rusanov_patch_update(KernelBuilder, 2, 4, 1, 5, 5, 1))
    Q2 = np.sin(3.141 * np.arange(360) / 360)
    p2.run(Q2, 1.0)
    assert np.array_equal(Q2[np.array(g["valid_modified_idx"])], np.array(g["valid_modified_val"]))


def test_hip_printer_aderdg_hint(exa, orc):
    from exahype_amd import KernelBuilder
    from exahype_amd.printers import HIPPrinter
    N, nc = 4, (2, 2, 2)
    k = KernelBuilder(3, N, 0, 5, 0, n_patches=8)
    k.item('u')
    u = euler_dg_state(nc + (N,) * 3, seed=8).reshape(8, N, N, N, 5)
    dx = [0.5] * 3
    want = orc.aderdg_step(u.reshape(-1), 1e-3, dx, _ops(N), 3, N, 5, orc.PDE_EULER, N, nc)
    HIPPrinter(k, scheme="aderdg", pde="euler").run(u, 1e-3, dx=dx)
    assert rel_err(u.reshape(-1), want) < TOL


# ---- BASELINE.json full size: size-independent properties ---------------------------------------------
def test_dg_full_size_conservation_128cubed(exa):
    """configs[2] (3-D Euler p=5, 128^3 cells): one full step conserves all five variables on the periodic
    grid to round-off and stays finite (KAT A.5-5 at full size; the oracle cannot run this in seconds)."""
    import torch
    N, nc = 6, (128, 128, 128)
    s = exa.AderDgSolver(3, N, nc)
    w = torch.as_tensor(s.operators()["w"], device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    sh = s.u.shape[:-1]
    for v, (base, amp) in enumerate(((1.0, 0.2), (0.5, 0.1), (-0.3, 0.1), (0.2, 0.1), (3.0, 0.2))):
        s.u[..., v] = base + amp * torch.rand(sh, generator=g, device="cuda", dtype=torch.float64)

    def mass():
        out = []
        for v in range(5):
            t = torch.einsum("abcijk,i,j,k->", s.u[..., v], w, w, w)
            out.append(float(t))
        return np.array(out)
    m0 = mass()
    s.step(2e-6)
    torch.cuda.synchronize()
    m1 = mass()
    assert bool(torch.isfinite(s.u).all())
    assert np.max(np.abs(m1 - m0) / np.abs(m0)) < 1e-12, (m0, m1)


def test_fv_full_size_replicated_golden_patch(exa, golden_dir):
    """SURVEY.md 8(d) FV-parity row at its full batch size (2^20 patches of the reference's configuration): every patch is
    the reference protocol's sin patch, so every patch must come out as the reference's golden vector -- bit for bit, the
    same in all 2^20 patches (a checksum of checksums: no patch slot, workgroup or XCD treats its data differently)."""
    import torch
    g = json.load(open(os.path.join(golden_dir, "fv_ref2d_sin.json")))
    n = 1 << 20
    one = torch.as_tensor(np.sin(3.141 * np.arange(360) / 360)).cuda()
    Q = one.repeat(n).reshape(n, 360)
    k = exa.FVRusanovKernel(2, 4, 1, 5, 5, n, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    k.time_step(Q.reshape(-1), 1.0)
    torch.cuda.synchronize()
    assert bool((Q == Q[0]).all())                                           # all patches identical ...
    first = Q[0].cpu().numpy()
    idx = np.array(g["valid_modified_idx"])
    assert np.array_equal(first[idx], np.array(g["valid_modified_val"]))     # ... and equal to the reference's output
    pidx = np.array(g["passthrough_idx"])
    assert np.array_equal(first[pidx], one.cpu().numpy()[pidx])


def test_dg_full_size_conservation_cfg1_512sq(exa):
    """configs[1] (2-D Euler p=3, 512 x 512 cells, single stage, fused launch): three steps conserve all variables to round-off."""
    import torch
    N, nc = 4, (512, 512)
    s = exa.AderDgSolver(2, N, nc, n_picard=0, fused_single_stage=True)
    assert s._fused
    w = torch.as_tensor(s.operators()["w"], device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    sh = s.u.shape[:-1]
    for v, (base, amp) in enumerate(((1.0, 0.2), (0.5, 0.1), (-0.3, 0.1), (0.0, 0.0), (3.0, 0.2))):
        s.u[..., v] = base + amp * torch.rand(sh, generator=g, device="cuda", dtype=torch.float64)

    def mass():
        return np.array([float(torch.einsum("abij,i,j->", s.u[..., v], w, w)) for v in range(5)])
    m0 = mass()
    for _ in range(3):
        s.step(1e-6)
    torch.cuda.synchronize()
    m1 = mass()
    assert bool(torch.isfinite(s.u).all())
    ok = np.abs(m0) > 0
    assert np.max(np.abs(m1 - m0)[ok] / np.abs(m0)[ok]) < 1e-12 and np.all(np.abs(m1[~ok]) < 1e-9), (m0, m1)


@pytest.mark.parametrize("N,nc", [(4, (5, 3)), (2, (4, 4)), (8, (2, 3)), (6, (7, 4)), (3, (1, 9)), (4, (130, 127)), (3, (190, 130)), (8, (100, 91))])
def test_dg_fused_single_stage_step_vs_oracle(exa, orc, N, nc):
    """Opt-in fused single-stage 2-D step (exa_dg_step_fused: traces stay on chip) == oracle; partial tiles, tiny grids, and grids with more tiles
    than workgroups fit on the chip (the persistent grid's tile loop with the next tile requested ahead; odd cell size: the 8-byte copy path)."""
    ops = _ops(N)
    u = euler_dg_state(tuple(nc) + (N, N), seed=900 + N)
    dx = [1.0 / c for c in nc]
    dt = 0.02 * min(dx) / (2 * N - 1)
    s = exa.AderDgSolver(2, N, nc, n_picard=0, dx=dx, fused_single_stage=True)
    assert s._fused
    s.upload(u)
    uo = u.reshape(-1).copy()
    for _ in range(3):
        s.step(dt)
        uo = orc.aderdg_step(uo, dt, dx, ops, 2, N, 5, orc.PDE_EULER, 0, nc)
    assert rel_err(s.download().reshape(-1), uo) < TOL
    # not offered where it does not apply
    assert not exa.AderDgSolver(2, N, nc, dx=dx, fused_single_stage=True)._fused
    assert not exa.AderDgSolver(3, 3, (2, 2, 2), n_picard=0, fused_single_stage=True)._fused


@pytest.mark.parametrize("p,meshes", [(2, (16, 32)), (3, (3, 6)), (5, (2, 4)), (7, (2, 4))])
def test_dg_order_of_accuracy_euler_density_wave_3d(exa, p, meshes):
    """KAT A.5-6 on the production kernels, against the ANALYTIC solution (independent of the oracle): the smooth Euler
    density wave rho = 1 + 0.2 sin(2 pi (x+y+z - 3t)), u = (1,1,1), p = 1 is pure advection; L2 error order >= p + 0.7."""
    from oracle import aderdg_numpy as A
    d, N = 3, p + 1
    ops = _ops(N)
    errs = []
    for nc in meshes:
        xs = A.node_coords((nc,) * d, N, ops)

        def exact(t):
            rho = 1 + 0.2 * np.sin(2 * np.pi * (xs[0] + xs[1] + xs[2] - d * t))
            q = np.zeros(rho.shape + (5,))
            q[..., 0] = rho
            for a in range(3):
                q[..., 1 + a] = rho
            q[..., 4] = 1 / 0.4 + 0.5 * rho * d
            return q
        s = exa.AderDgSolver(d, N, (nc,) * d)
        s.upload(exact(0.0))
        T = 0.04
        s.run(T, cfl=0.3)
        w = ops["w"]
        err2 = np.einsum("abcijk,i,j,k->", (s.download() - exact(T))[..., 0] ** 2, w, w, w) / nc ** d
        errs.append(np.sqrt(err2))
    order = np.log2(errs[0] / errs[1])
    assert order >= p + 0.7, (errs, order)
