"""Known-answer tests of SURVEY.md A.5 -- they stand in for the golden vectors the reference does not
have for ADER-DG ("parity unpinned") and pin the oracle the GPU kernels are checked
against."""
import numpy as np
import pytest

import oracle
from oracle import aderdg_numpy as A
from oracle.dg_operators import operators


@pytest.mark.parametrize("N", range(2, 9))
def test_operator_identities(N):
    o = operators(N)
    D, K, w, pL, pR, iK1, F0 = o["D"], o["Kxi"], o["w"], o["phiL"], o["phiR"], o["iK1"], o["F0"]
    assert abs(w.sum() - 1) < 1e-15
    assert np.max(np.abs(D.sum(1))) < 1e-13                                              # D 1 = 0
    assert np.max(np.abs(K + K.T - np.outer(pR, pR) + np.outer(pL, pL))) < 1e-13          # integration by parts
    assert np.max(np.abs(K.sum(1) - (pR - pL))) < 1e-13 and np.max(np.abs(K.sum(0))) < 1e-14
    assert np.max(np.abs(iK1 @ F0 - 1)) < 1e-13                                          # predictor preserves constants
    assert np.linalg.cond(o["K1"]) < 20


def test_constant_state_is_fixed_point():
    ops = operators(4)
    u = np.ones((3, 3, 3, 4, 4, 4, 5)) * np.array([1.2, 0.3, -0.2, 0.5, 2.5])
    assert np.max(np.abs(A.step(u, 0.01, [1 / 3] * 3, ops, A.Euler()) - u)) < 1e-13
    un = oracle.aderdg_step(u.reshape(-1), 0.01, [1 / 3] * 3, ops, 3, 4, 5, oracle.PDE_EULER, 4, (3, 3, 3))
    assert np.max(np.abs(un - u.reshape(-1))) < 1e-13


def test_picard_terminates_for_linear_flux():
    N = 6
    ops = operators(N)
    xs = A.node_coords((4,), N, ops)
    u = np.sin(2 * np.pi * xs[0])[..., None]
    pde = A.Advection([1.0])
    qN = A.predictor(u, 0.02, [0.25], ops, pde, n_it=N)
    qN1 = A.predictor(u, 0.02, [0.25], ops, pde, n_it=N + 1)
    assert np.max(np.abs(qN1 - qN)) < 1e-12


@pytest.mark.parametrize("dim", [1, 2])
def test_exact_on_global_polynomial(dim):
    N, nc = 4, (6,) * dim
    ops = operators(N)
    xs = A.node_coords(nc, N, ops)
    a = [1.0, 0.5][:dim]
    poly = (lambda x: 1 + x - x ** 2 + 2 * x ** 3) if dim == 1 else (lambda x, y: 1 + x - 2 * y + x * y + x ** 3 - y ** 2 * x)
    u = (poly(*xs) * np.ones(nc + (N,) * dim))[..., None]
    dt = 0.02
    un = A.step(u, dt, [1 / 6] * dim, ops, A.Advection(a))[..., 0]
    want = poly(*[xs[d] - a[d] * dt for d in range(dim)]) * np.ones_like(un)
    inner = (slice(1, -1),) * dim
    assert np.max(np.abs(un[inner] - want[inner])) < 1e-12


def test_conservation_euler_3d():
    N, nc = 3, (3, 3, 3)
    ops = operators(N)
    from tests.util import euler_dg_state
    u = euler_dg_state(nc + (N,) * 3, seed=3)
    W = np.einsum("i,j,k->ijk", ops["w"], ops["w"], ops["w"])[..., None]
    m0 = (u * W).sum(axis=tuple(range(6)))
    for _ in range(4):
        u = A.step(u, 2e-3, [1 / 3] * 3, ops, A.Euler())
    m1 = (u * W).sum(axis=tuple(range(6)))
    assert np.max(np.abs(m1 - m0) / np.abs(m0).max()) < 1e-13


@pytest.mark.parametrize("p", [1, 3, 5])
def test_order_of_accuracy_advection_1d(p):
    N = p + 1
    ops = operators(N)
    errs = []
    for nc in (4, 8, 16):
        xs = A.node_coords((nc,), N, ops)
        u = np.sin(2 * np.pi * xs[0])[..., None]
        h = 1 / nc
        T = 0.1
        ns = int(np.ceil(T / (0.2 * h / (2 * p + 1))))
        for _ in range(ns):
            u = A.step(u, T / ns, [h], ops, A.Advection([1.0]))
        ex = np.sin(2 * np.pi * (xs[0] - T))[..., None]
        errs.append(np.sqrt(((u - ex) ** 2 * ops["w"][None, :, None]).sum() * h))
    order = np.log2(errs[1] / errs[2])
    assert order >= p + 0.7, (errs, order)


def test_order_of_accuracy_euler_density_wave_2d():
    p, d = 3, 2
    N = p + 1
    ops = operators(N)
    errs = []
    for nc in (4, 8):
        xs = A.node_coords((nc,) * d, N, ops)

        def exact(t):
            s = xs[0] + xs[1] - d * t
            rho = 1 + 0.2 * np.sin(2 * np.pi * s)
            q = np.zeros(np.broadcast(rho).shape + (5,))
            q[..., 0] = rho; q[..., 1] = rho; q[..., 2] = rho
            q[..., 4] = 1 / 0.4 + 0.5 * rho * 2
            return q
        u = exact(0.0)
        h = 1 / nc
        T = 0.05
        ns = int(np.ceil(T / (0.3 * h / ((2 * p + 1) * d * 2.4))))
        for _ in range(ns):
            u = A.step(u, T / ns, [h] * d, ops, A.Euler())
        errs.append(np.sqrt(((u - exact(T))[..., 0] ** 2).mean()))
    assert np.log2(errs[0] / errs[1]) >= p + 0.7, errs
