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


def reaction_advection(k=3.0, max_dim=3):
    """Two advected species with a linear reaction chain -- a system with an algebraic source term:
    q_t + a . grad q = S(q),  S = (-k q0, k q0 - 2 k q1)."""
    from exahype_amd.pde_codegen import SympyPDE
    a = (1.0, 0.5, -0.25)
    return SympyPDE(2, flux=lambda q, d: [a[d] * q[0], a[d] * q[1]], max_eigenvalue=lambda q, d: sympy.Float(abs(a[d])),
                    source=lambda q: [-k * q[0], k * q[0] - 2 * k * q[1]], max_dim=max_dim, name="reaction_advection")


class NumpyPDE:
    """The same expressions, lambdified, with the interface oracle/aderdg_numpy.py expects."""

    def __init__(self, spde):
        self.m = spde.n_vars
        self._f = [sympy.lambdify(spde.q, spde.flux_exprs[d], "numpy") for d in range(spde.max_dim)]
        self._e = [sympy.lambdify(spde.q, spde.eig_exprs[d], "numpy") for d in range(spde.max_dim)]
        if getattr(spde, "source_exprs", None) is not None:
            s_ = sympy.lambdify(spde.q, spde.source_exprs, "numpy")
            self.source = lambda q: np.stack([np.broadcast_to(o, q.shape[:-1]) for o in s_(*[q[..., v] for v in range(self.m)])], axis=-1)

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


def test_source_term_reaches_the_generated_device_code():
    p = reaction_advection()
    src = p.source()
    assert "HAS_SOURCE = true" in src and "static inline void source(const double* q, double* S)" in src
    # and through the operator surface: a function named like the reference harness's hook, with a body
    from exahype_amd import KernelBuilder
    from exahype_amd.printers import HIPPrinter
    k = KernelBuilder(dim=2, patch_size=4, halo_size=0, n_real=2, n_aux=0, n_patches=6)
    k.item('u')
    k.function('flux', body=lambda q, d: [q[0], q[1]])
    k.function('maxEigenvalue', body=lambda q, d: sympy.Integer(1))
    k.function('sourceTerm', body=lambda q: [-3 * q[0], 3 * q[0] - 6 * q[1]])
    hp = HIPPrinter(k, scheme="aderdg", grid=(3, 2))
    assert "HAS_SOURCE = true" in hp.user_pde.source() and "S[1] = -6*q[1] + t_0;" in hp.user_pde.source()   # (common sub-expression 3 q0)
    assert "HAS_SOURCE" not in swe().source()                   # term sets without a source generate none
    with pytest.raises(ValueError):
        from exahype_amd.pde_codegen import SympyPDE
        SympyPDE(2, flux=lambda q, d: [q[0], q[1]], max_eigenvalue=lambda q, d: 1, source=lambda q: [q[0]])


@pytest.mark.gpu
@pytest.mark.parametrize("dim,N,nc", [(2, 4, (3, 2)), (3, 6, (2, 1, 2)), (3, 5, (2, 2, 1)), (2, 3, (2, 3)),
                                       (3, 8, (1, 2, 1)), (3, 7, (2, 1, 1))])         # level-streamed kernel (p = 7, 6)
def test_source_term_aderdg_vs_numpy_oracle(dim, N, nc):
    """q_t + div F = S(q): the source enters the predictor beside the flux divergence, its time average the volume update
    (oracle/aderdg_numpy.py).  Full predictor and single-stage scheme; both forms of the Picard loop (N = 6 / N = 5, 3) and the
    level-streamed kernel (N = 8, 7)."""
    from exahype_amd import solvers as exa
    from oracle import aderdg_numpy as A
    from oracle.dg_operators import operators
    p = reaction_advection()
    rng = np.random.default_rng(N)
    u = 1.0 + 0.3 * rng.random(tuple(nc) + (N,) * dim + (2,))
    dx = [1.0 / c for c in nc]
    dt = 0.05 * min(dx) / (2 * N - 1)
    for n_picard, ref_step in ((-1, lambda v: A.step(v, dt, dx, operators(N), NumpyPDE(p))),
                               (0, lambda v: A.step_single_stage(v, dt, dx, operators(N), NumpyPDE(p)))):
        s = exa.AderDgSolver(dim, N, nc, pde=p.register(), n_vars=2, dx=dx, n_picard=n_picard)
        s.upload(u)
        ref = u.copy()
        for _ in range(2):
            s.step(dt)
            ref = ref_step(ref)
        assert np.max(np.abs(s.download() - ref)) / np.max(np.abs(ref)) < 1e-10, n_picard
    # the oracle WITHOUT the source is far from what the kernel produced: the term is really in the kernel
    plain = NumpyPDE(p)
    del plain.source
    ref0 = u.copy()
    for _ in range(2):
        ref0 = A.step_single_stage(ref0, dt, dx, operators(N), plain)
    assert np.max(np.abs(s.download() - ref0)) > 1e-6


@pytest.mark.gpu
def test_source_term_decay_matches_the_exact_solution():
    """A constant state obeys the ODE q' = S(q): q0 = e^{-kt}, q1 = e^{-kt} - e^{-2kt} from (1, 0).  ADER-DG of order N integrates
    it to ~ (k dt)^(N+1) per step."""
    from exahype_amd import solvers as exa
    k, N, nc = 3.0, 6, (2, 2, 2)
    p = reaction_advection(k)
    u = np.zeros(nc + (N,) * 3 + (2,))
    u[..., 0] = 1.0
    s = exa.AderDgSolver(3, N, nc, pde=p.register(), n_vars=2, dx=[0.5] * 3)
    s.upload(u)
    T, steps = 0.2, 8
    for _ in range(steps):
        s.step(T / steps)
    got = s.download()
    exact = np.array([np.exp(-k * T), np.exp(-k * T) - np.exp(-2 * k * T)])
    assert np.max(np.abs(got - exact)) < 1e-9
    assert np.max(np.abs(got - got[0, 0, 0, 0, 0, 0])) < 1e-10          # and stays constant in space (D applied to a constant is round-off)


@pytest.mark.gpu
def test_source_term_fv_rusanov_vs_numpy():
    from exahype_amd import solvers as exa
    p = reaction_advection()
    npde = NumpyPDE(p)
    n_patches, P, H = 4, 5, 1
    S = P + 2 * H
    Q = 1.0 + 0.3 * np.random.default_rng(2).random((n_patches, S, S, 2))
    dt, h = 2e-3, 0.05
    kern = exa.FVRusanovKernel(2, P, H, 2, 0, n_patches, pde=p.register(), mode=exa.FV_RUSANOV)
    got = np.ascontiguousarray(Q.copy())
    kern.time_step(got, dt, h)
    acc = np.zeros((n_patches, P, P, 2))
    core = (slice(None), slice(H, H + P), slice(H, H + P))
    for d in range(2):
        sh = lambda a, s_: np.roll(a, -s_, axis=1 + d)[core]
        qc, qp, qm = Q[core], sh(Q, 1), sh(Q, -1)
        lc, lp, lm = npde.maxeig(qc, d), npde.maxeig(qp, d), npde.maxeig(qm, d)
        Fc, Fp, Fm = npde.flux(qc, d), npde.flux(qp, d), npde.flux(qm, d)
        acc += 0.5 * (Fc + Fp) - 0.5 * np.maximum(lc, lp)[..., None] * (qp - qc)
        acc -= 0.5 * (Fm + Fc) - 0.5 * np.maximum(lm, lc)[..., None] * (qc - qm)
    want = Q.copy()
    want[core] = Q[core] - dt / h * acc + dt * npde.source(Q[core])
    assert np.max(np.abs(got - want)) < 1e-12


