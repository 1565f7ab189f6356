"""PDE terms given as SymPy expressions, compiled for the device and registered at run time.
CPU part: code generation, JIT cross-compile, registration.  GPU part: a shallow-water system (not built in)
against numpy restatements driven by the SAME SymPy expressions (lambdified), and Euler-from-SymPy against the
built-in Euler terms."""
import numpy as np
import pytest
import sympy

G = 9.81


def swe():
    from exahype_amd.pde_codegen import SympyPDE

    def flux(q, d):
        h, hu, hv = q
        un = (hu, hv)[d] / h if d < 2 else 0
        p = sympy.Rational(1, 2) * G * h * h
        f = [h * un, hu * un, hv * un]
        if d < 2:
            f[1 + d] = f[1 + d] + p
        return f

    def eig(q, d):
        h, hu, hv = q
        un = (hu, hv)[d] / h if d < 2 else 0
        return sympy.Abs(un) + sympy.sqrt(G * h)
    return SympyPDE(3, flux, eig, max_dim=2, name="shallow_water")


def euler_sympy():
    from exahype_amd.pde_codegen import SympyPDE

    def prim(q):
        irho = 1 / q[0]
        p = sympy.Float(0.4) * (q[4] - sympy.Rational(1, 2) * irho * (q[1] ** 2 + q[2] ** 2 + q[3] ** 2))
        return irho, p

    def flux(q, d):
        irho, p = prim(q)
        c = irho * q[d + 1]
        f = [c * q[0], c * q[1], c * q[2], c * q[3], c * q[4] + c * p]
        f[d + 1] = f[d + 1] + p
        return f

    def eig(q, d):
        irho, p = prim(q)
        return sympy.Abs(q[d + 1] * irho) + sympy.sqrt(sympy.Float(1.4) * p * irho)
    return SympyPDE(5, flux, eig, max_dim=3, name="euler_from_sympy")


class NumpyPDE:
    """The same expressions, lambdified, with the interface oracle/aderdg_numpy.py expects."""

    def __init__(self, spde):
        self.m = spde.n_vars
        self._f = [sympy.lambdify(spde.q, spde.flux_exprs[d], "numpy") for d in range(spde.max_dim)]
        self._e = [sympy.lambdify(spde.q, spde.eig_exprs[d], "numpy") for d in range(spde.max_dim)]

    def flux(self, q, d):
        out = self._f[d](*[q[..., v] for v in range(self.m)])
        return np.stack([np.broadcast_to(o, q.shape[:-1]) for o in out], axis=-1)

    def maxeig(self, q, d):
        return np.broadcast_to(self._e[d](*[q[..., v] for v in range(self.m)]), q.shape[:-1])


def test_codegen_and_registration_without_gpu():
    from exahype_amd import _lib
    p = swe()
    src = p.source()
    assert "struct UserPDE" in src and "NV = 3" in src and "MAXDIM = 2" in src and "sqrt(" in src and "fabs(" in src
    assert p.key() == swe().key()                       # content-addressed
    so = p.build()
    import os
    assert os.path.exists(so)
    pid = p.register()
    assert pid >= 100 and p.register() == pid


def swe_state(shape, seed):
    rng = np.random.default_rng(seed)
    q = np.zeros(tuple(shape) + (3,))
    q[..., 0] = 1.0 + 0.3 * rng.random(shape)
    q[..., 1] = q[..., 0] * (0.4 * rng.random(shape) - 0.2)
    q[..., 2] = q[..., 0] * (0.4 * rng.random(shape) - 0.2)
    return q


@pytest.mark.gpu
def test_user_pde_aderdg_vs_numpy_oracle():
    from exahype_amd import solvers as exa
    from oracle import aderdg_numpy as A
    from oracle.dg_operators import operators
    p = swe()
    N, nc = 4, (4, 3)
    u = swe_state(nc + (N, N), 3)
    dx = [1.0 / c for c in nc]
    dt = 2e-3
    s = exa.AderDgSolver(2, N, nc, pde=p.register(), n_vars=3, dx=dx)
    s.upload(u)
    ref = u.copy()
    for _ in range(3):
        s.step(dt)
        ref = A.step(ref, dt, dx, operators(N), NumpyPDE(p))
    got = s.download()
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-10


@pytest.mark.gpu
def test_user_pde_fv_rusanov_vs_numpy():
    from exahype_amd import solvers as exa
    p = swe()
    npde = NumpyPDE(p)
    n_patches, P, H = 5, 6, 1
    S = P + 2 * H
    Q = swe_state((n_patches, S, S), 9)
    dt, h = 1e-3, 0.05
    k = exa.FVRusanovKernel(2, P, H, 3, 0, n_patches, pde=p.register(), mode=exa.FV_RUSANOV)
    got = np.ascontiguousarray(Q.copy())
    k.time_step(got, dt, h)
    # corrected Rusanov in numpy with the lambdified terms
    want = Q.copy()
    acc = np.zeros((n_patches, P, P, 3))
    core = (slice(None), slice(H, H + P), slice(H, H + P))
    for d in range(2):
        sh = lambda a, s_: np.roll(a, -s_, axis=1 + d)[core]
        qc, qp, qm = Q[core], sh(Q, 1), sh(Q, -1)
        lc, lp, lm = npde.maxeig(qc, d), npde.maxeig(qp, d), npde.maxeig(qm, d)
        Fc, Fp, Fm = npde.flux(qc, d), npde.flux(qp, d), npde.flux(qm, d)
        acc += 0.5 * (Fc + Fp) - 0.5 * np.maximum(lc, lp)[..., None] * (qp - qc)
        acc -= 0.5 * (Fm + Fc) - 0.5 * np.maximum(lm, lc)[..., None] * (qc - qm)
    want[core] = Q[core] - dt / h * acc
    assert np.max(np.abs(got - want)) < 1e-12


@pytest.mark.gpu
def test_euler_from_sympy_equals_builtin_euler():
    from exahype_amd import solvers as exa
    from tests.util import euler_dg_state
    p = euler_sympy()
    N, nc = 3, (2, 2, 2)
    u = euler_dg_state(nc + (N,) * 3, seed=21)
    a = exa.AderDgSolver(3, N, nc, pde=exa.PDE_EULER)
    b = exa.AderDgSolver(3, N, nc, pde=p.register())
    a.upload(u); b.upload(u)
    for _ in range(2):
        a.step(1e-3); b.step(1e-3)
    assert np.max(np.abs(a.download() - b.download())) / np.max(np.abs(u)) < 1e-12
    Q = u.reshape(-1, 5)
    for d in range(3):
        Fa, la = exa.pde_eval(exa.PDE_EULER, d, Q)
        Fb, lb = exa.pde_eval(p.register(), d, Q)
        assert np.max(np.abs(Fa - Fb)) < 1e-13 and np.max(np.abs(la - lb)) < 1e-13


def _swe_kernel(n_patches, patch_size=4, halo_size=0):
    """The same shallow-water terms handed over through the operator surface: kernel.function(..., body=...)."""
    from exahype_amd import KernelBuilder
    p = swe()
    k = KernelBuilder(dim=2, patch_size=patch_size, halo_size=halo_size, n_real=3, n_aux=0, n_patches=n_patches)
    k.item('u')
    k.function('Flux', parameter_types=['double*', 'int', 'double*'], return_type='void',
               body=lambda q, d: p.flux_exprs[d] if q == p.q else [e.subs(dict(zip(p.q, q))) for e in p.flux_exprs[d]])
    k.function('maxEigenvalue', parameter_types=['double*', 'int'], return_type='double',
               body=lambda q, d: p.eig_exprs[d].subs(dict(zip(p.q, q))))
    return k, p


def test_function_bodies_reach_the_printer_without_gpu():
    from exahype_amd import KernelBuilder
    from exahype_amd.printers import HIPPrinter
    k, p = _swe_kernel(6)
    hp = HIPPrinter(k, scheme="aderdg", grid=(3, 2))
    assert hp.user_pde is not None and hp.user_pde.n_vars == 3
    body = lambda src: src.split("\n", 1)[1]            # (the first line carries the term set's name)
    assert body(hp.user_pde.source()) == body(swe().source())   # same expressions -> same device code as the SympyPDE route
    # a body for one of the two terms only is an error; no bodies: the built-in term sets as before
    k2 = KernelBuilder(dim=2, patch_size=4, halo_size=0, n_real=3, n_aux=0, n_patches=6)
    k2.item('u')
    k2.function('Flux', body=lambda q, d: [q[0]] * 3)
    with pytest.raises(ValueError):
        HIPPrinter(k2, scheme="aderdg", grid=(3, 2))
    with pytest.raises(TypeError):
        k2.function('maxEigenvalue', body=3.0)


@pytest.mark.gpu
def test_function_bodies_run_like_the_sympy_pde():
    from exahype_amd.printers import HIPPrinter
    N, grid = 4, (3, 2)
    k, p = _swe_kernel(6, patch_size=N)
    rng = np.random.default_rng(3)
    u = np.empty(grid + (N, N, 3))
    u[..., 0] = 1.0 + 0.2 * rng.random(grid + (N, N))
    u[..., 1:] = 0.1 * (rng.random(grid + (N, N, 2)) - 0.5)
    dx = (1.0 / 3, 0.5)
    a = u.copy()
    HIPPrinter(k, scheme="aderdg", grid=grid).run(a, 1e-3, dx=dx, steps=2)
    from exahype_amd import KernelBuilder
    k0 = KernelBuilder(dim=2, patch_size=N, halo_size=0, n_real=3, n_aux=0, n_patches=6)
    k0.item('u')
    b = u.copy()
    HIPPrinter(k0, scheme="aderdg", grid=grid, pde=swe()).run(b, 1e-3, dx=dx, steps=2)
    assert np.array_equal(a, b) and not np.array_equal(a, u)
