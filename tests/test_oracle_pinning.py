"""Pin the CPU oracle (oracle/exa_oracle.c) before trusting it: against the golden vectors captured
from the compiled reference (tests/golden/*.json), against the compiled reference itself when
oracle/_ref is present, and (for the parts the reference does not have) against the independent
numpy restatement."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import aderdg_numpy as A
from oracle.dg_operators import operators
from tests.util import euler_dg_state, euler_ref2d_patches


def test_fv_faithful_anchor_values_and_mask(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "fv_ref2d_sin.json")))
    Q = np.sin(3.141 * np.arange(360) / 360)
    out = oracle.fv_faithful(Q, 1.0, 2, 4, 1, 5, 5)
    idx = np.array(g["valid_modified_idx"])
    assert idx.tolist() == [360 * 0 + 60 * i + 10 * j + v for i in (2, 3) for j in (2, 3) for v in range(4)]
    assert np.array_equal(out[idx], np.array(g["valid_modified_val"]))
    # SURVEY.md 8(c) anchor values, printed by the survey from the g++ build of the reference
    for i, v in ((140, 0.53869414698947504), (141, 0.75203086118104856), (153, 0.84088465449056549), (213, 1.1310494339105341)):
        assert out[i] == v
    p = np.array(g["passthrough_idx"])
    assert np.array_equal(out[p], Q[p])


def test_fv_faithful_random_patches_vs_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "fv_ref2d_random.json")))
    Q = np.array(g["Q_in"])
    out = oracle.fv_faithful(Q, g["dt"], 2, 4, 1, 5, 5, n_patches=Q.shape[0])
    assert np.array_equal(out[:, 2:4, 2:4, 0:4], np.array(g["Q_out_valid"]))


def test_euler_terms_vs_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "euler_terms_ref2d.json")))
    Q = np.array(g["Q"]); fl = np.array(g["flux"]); ev = np.array(g["maxeig"])
    L = oracle.lib()
    for i in range(len(Q)):
        for d in range(2):
            F = np.zeros(5)
            L.orc_pde_flux(oracle.PDE_EULER_REF2D, 5, np.ascontiguousarray(Q[i]), d, F)
            assert np.array_equal(F[:4], fl[i, d]) and F[4] == 0.0
            assert L.orc_pde_maxeig(oracle.PDE_EULER_REF2D, np.ascontiguousarray(Q[i]), d) == ev[i, d]


@pytest.mark.skipif(oracle.ref() is None, reason="oracle/_ref not built (no /root/reference on this machine)")
def test_fv_faithful_vs_compiled_reference_live():
    R = oracle.ref()
    Q = euler_ref2d_patches(16, 6, 10, seed=11)
    want = Q.copy()
    for p in range(16):
        tmp = np.ascontiguousarray(want[p]).ravel()
        R.ref_time_step(tmp, 0.21)
        want[p] = tmp.reshape(6, 6, 10)
    got = oracle.fv_faithful(Q, 0.21, 2, 4, 1, 5, 5, n_patches=16)
    assert np.array_equal(got[:, 2:4, 2:4, :4], want[:, 2:4, 2:4, :4])       # defined outputs only
    assert np.array_equal(got[..., 5:], want[..., 5:])                       # aux variables pass through


def test_fv_corrected_properties():
    """Unpinned by the reference (it has no correct Rusanov): constant state is a fixed point, the halo and
    aux variables are untouched, and the update is conservative for a periodic patch."""
    dim, P, H = 2, 8, 1
    S = P + 2 * H
    Q = np.zeros((1, S, S, 7)); Q[..., :5] = [1.1, 0.3, -0.2, 0.1, 2.6]; Q[..., 5:] = 7.0
    out = oracle.fv_corrected(Q, 0.01, 0.1, dim, P, H, 5, 2, pde=oracle.PDE_EULER)
    assert np.max(np.abs(out - Q)) < 1e-15
    rng = np.random.default_rng(0)
    Q = np.zeros((1, S, S, 5)); core = euler_dg_state((P, P), 4, amp=0.3)
    Q[0, H:-H, H:-H] = core
    Q[0, 0] = Q[0, -2]; Q[0, -1] = Q[0, 1]; Q[0, :, 0] = Q[0, :, -2]; Q[0, :, -1] = Q[0, :, 1]       # periodic halo
    out = oracle.fv_corrected(Q, 0.002, 0.1, dim, P, H, 5, 0, pde=oracle.PDE_EULER)
    assert np.max(np.abs(out[0, H:-H, H:-H].sum(axis=(0, 1)) - core.sum(axis=(0, 1)))) < 1e-12
    assert np.array_equal(out[0, 0], Q[0, 0]) and np.array_equal(out[0, :, -1], Q[0, :, -1])


@pytest.mark.parametrize("dim,N,nc", [(2, 4, (3, 2)), (3, 3, (2, 3, 2)), (3, 6, (2, 2, 2))])
def test_aderdg_c_oracle_equals_numpy_restatement(dim, N, nc):
    ops = operators(N)
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=N)
    dx = [1.0 / c for c in nc]
    dt = 0.01 * min(dx)
    st = A.step(u, dt, dx, ops, A.Euler(), stages=True)
    us, tr = oracle.aderdg_stage_a(u.reshape(-1), dt, dx, ops, dim, N, 5, oracle.PDE_EULER, N)
    assert np.max(np.abs(us.reshape(u.shape) - st["ustar"])) < 1e-13
    for a in range(dim):
        qL, qR, FL, FR = st["traces"][a]
        for side, (qq, FF) in enumerate(((qL, FL), (qR, FR))):
            want = np.stack([np.moveaxis(qq, -1, dim).reshape(int(np.prod(nc)), 5, -1),
                             np.moveaxis(FF, -1, dim).reshape(int(np.prod(nc)), 5, -1)], axis=1)
            assert np.max(np.abs(tr[a, side] - want)) < 1e-13
    un = oracle.aderdg_step(u.reshape(-1), dt, dx, ops, dim, N, 5, oracle.PDE_EULER, N, nc)
    assert np.max(np.abs(un.reshape(u.shape) - st["unew"])) < 1e-13
    q = oracle.aderdg_predictor(u[(0,) * dim], dt, dx, ops, dim, N, 5, oracle.PDE_EULER, N)
    assert np.max(np.abs(q - st["q"][(0,) * dim])) < 1e-13
    u1 = oracle.aderdg_step(u.reshape(-1), dt, dx, ops, dim, N, 5, oracle.PDE_EULER, 0, nc)
    assert np.max(np.abs(u1.reshape(u.shape) - A.step_single_stage(u, dt, dx, ops, A.Euler()))) < 1e-13
