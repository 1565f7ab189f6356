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
@pytest.mark.parametrize("N", [3, 6, 8])
def test_euler_from_sympy_equals_builtin_euler(N):
    """The north-star's drop-in sentence on the kernels that carry the benchmarks: N = 6 is the register-resident stage A (generated
    Dir / flux_scaled_dir), N = 8 the matrix-pipe one (generated flux_scaled + cached scalars), N = 3 the generic LDS kernel."""
    from exahype_amd import solvers as exa
    from tests.util import euler_dg_state
    p = euler_sympy()
    nc = (2, 2, 2) if N < 8 else (2, 1, 2)
    u = euler_dg_state(nc + (N,) * 3, seed=21)
    a = exa.AderDgSolver(3, N, nc, pde=exa.PDE_EULER)
    b = exa.AderDgSolver(3, N, nc, pde=p.register())
    if N == 6:
        assert "reg_kernel" in a.stage_a_kernel_name() and "reg_kernel" in b.stage_a_kernel_name() and "UserPDE" in b.stage_a_kernel_name()
    if N == 8:
        assert "m8_kernel" in a.stage_a_kernel_name() and "m8_kernel" in b.stage_a_kernel_name()
    a.upload(u); b.upload(u)
    for _ in range(2):
        a.step(1e-3); b.step(1e-3)
    assert np.max(np.abs(a.download() - b.download())) / np.max(np.abs(u)) < 1e-12
    Q = u.reshape(-1, 5)
    for d in range(3):
        Fa, la = exa.pde_eval(exa.PDE_EULER, d, Q)
        Fb, lb = exa.pde_eval(p.register(), d, Q)
        assert np.max(np.abs(Fa - Fb)) < 1e-13 and np.max(np.abs(la - lb)) < 1e-13


def test_generated_term_set_caches_shared_scalars_and_masks_the_direction():
    """pde_codegen._analyse: Euler written the obvious way gets the two cached scalars of the hand-written exa::Euler (1/rho, p), fast
    reciprocals in the members the ADER-DG kernels call (IEEE in flux_rt / maxeig: the FV path), and the per-lane-normal flux as
    straight-line code; a system with nothing to share gets no LDS slot."""
    p = euler_sympy()
    src = p.source()
    assert p.n_aux() == 2 and "NAUX = 2" in src
    aux = [sympy.simplify(e) for e in p._aux]
    q = p.q
    assert sympy.simplify(aux[0] - 1 / q[0]) == 0
    assert sympy.simplify(aux[1] - sympy.Float(0.4) * (q[4] - (q[1] ** 2 + q[2] ** 2 + q[3] ** 2) / (2 * q[0]))) == 0
    assert "struct Dir" in src and "flux_scaled_dir" in src and p.dir_form["masked_ops"] <= 1.8 * p.dir_form["one_direction_ops"]
    body = lambda name: src[src.index(name):src.index("}", src.index(name))]
    assert "exa::fast_rcp" in body("void aux_fast(") and "1.0/" in body("void aux(")
    tuned = src[src.index("template <int D> __device__ static inline void flux("):src.index("__device__ static inline double maxeig(")]
    assert "/" not in tuned.replace("//", "")                         # no IEEE division where the ADER-DG kernels evaluate the flux
    assert "exa::fast_sqrt" in src[src.index("maxeig_fast("):] and "sqrt(" in body("double maxeig(")
    # the benchmark's own statement of the system (bench.sympy_euler: u_n (E + p), quotients instead of a named 1/rho) ends at the same two scalars
    import bench
    b = bench.sympy_euler()
    assert b.n_aux() == 2 and sympy.simplify(b._aux[0] - aux[0]) == 0 and sympy.simplify(b._aux[1] - aux[1]) == 0 and "struct Dir" in b.source()
    r = reaction_advection()
    assert r.n_aux() == 0 and "NAUX = 0" in r.source()
    w = swe()
    assert w.n_aux() == 1 and sympy.simplify(w._aux[0] - 1 / w.q[0]) == 0 and "struct Dir" not in w.source()     # (2-D only: no per-lane normal)
    assert euler_sympy().__class__(5, lambda q, d: euler_sympy().flux_exprs[d], lambda q, d: euler_sympy().eig_exprs[d], max_aux=0).n_aux() == 0


def test_generated_flux_forms_agree_with_the_expressions():
    """Every generated form of the flux (cached scalars, scaled, direction-masked) is the user's expression: evaluated symbolically at
    random states."""
    p = euler_sympy()
    p._analyse()
    rng = np.random.default_rng(5)
    A = p._aux_syms
    for _ in range(3):
        qv = {s: float(v) for s, v in zip(p.q, [1.0 + rng.random(), rng.random() - 0.5, rng.random() - 0.5, rng.random() - 0.5, 3.0 + rng.random()])}
        av = {a: e.xreplace(qv) for a, e in zip(A, p._aux)}
        for d in range(3):
            for v in range(5):
                want = float(p.flux_exprs[d][v].xreplace(qv))
                assert abs(float(p._flux_a[d][v].xreplace(av).xreplace(qv)) - want) < 1e-13
        nn = list(sympy.symbols("n0:3"))
        masked = p._best_form([sum(nn[d] * p._flux_a[d][v] for d in range(3)) for v in range(5)], A + nn)
        for d in range(3):
            nv = {nn[k]: (0.7 if k == d else 0.0) for k in range(3)}
            for v in range(5):
                assert abs(float(masked[v].xreplace(nv).xreplace(av).xreplace(qv)) - 0.7 * float(p.flux_exprs[d][v].xreplace(qv))) < 1e-13


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




def euler_gravity(g=(0.3, -0.5, 0.8)):
    """Compressible Euler with a constant body force: a NONLINEAR five-variable system with a source, S = (0, rho g, m . g)."""
    from exahype_amd.pde_codegen import SympyPDE
    base = euler_sympy()
    q = base.q
    return SympyPDE(5, flux=lambda qq, d: [e.subs(dict(zip(q, qq))) for e in base.flux_exprs[d]],
                    max_eigenvalue=lambda qq, d: base.eig_exprs[d].subs(dict(zip(q, qq))),
                    source=lambda qq: [0, qq[0] * g[0], qq[0] * g[1], qq[0] * g[2], qq[1] * g[0] + qq[2] * g[1] + qq[3] * g[2]],
                    max_dim=3, name="euler_gravity")


@pytest.mark.gpu
@pytest.mark.parametrize("N,nc", [(6, (2, 2, 1)), (8, (1, 1, 2)), (4, (2, 1, 2))])
def test_nonlinear_five_variable_source_aderdg_vs_numpy_oracle(N, nc):
    """Euler + gravity through every 3-D stage-A kernel that carries the source hooks: the register-resident kernel (N = 6; five variables: the
    same LDS image and lane tables as the built-in Euler, flux of a generated term set dispatched per lane), the matrix-pipe kernel (N = 8) and
    the LDS-resident kernel (N = 4).  Parity unpinned against the reference (no source fixture there): oracle/aderdg_numpy.py with the same
    lambdified expressions."""
    from exahype_amd import solvers as exa
    from oracle import aderdg_numpy as A
    from oracle.dg_operators import operators
    from tests.util import euler_dg_state
    p = euler_gravity()
    u = euler_dg_state(tuple(nc) + (N,) * 3, seed=60 + N)
    dx = [1.0 / c for c in nc]
    dt = 0.02 * min(dx) / (2 * N - 1)
    s = exa.AderDgSolver(3, N, nc, pde=p.register(), n_vars=5, dx=dx)
    s.upload(u)
    ref = u.copy()
    for _ in range(2):
        s.step(dt)
        ref = A.step(ref, dt, dx, operators(N), NumpyPDE(p))
    assert np.max(np.abs(s.download() - ref)) / np.max(np.abs(ref)) < 1e-10
    plain = NumpyPDE(p)
    del plain.source
    ref0 = u.copy()
    for _ in range(2):
        ref0 = A.step(ref0, dt, dx, operators(N), plain)
    assert np.max(np.abs(s.download() - ref0)) > 1e-7


@pytest.mark.gpu
def test_source_term_fv_rusanov_3d_plane_streaming_vs_numpy():
    """3-D patches of 15^3 volumes (the limiter's patch at p = 7: plane-streaming kernel, non-cached path for generated term sets) with a source."""
    from exahype_amd import solvers as exa
    from tests.util import euler_dg_state
    p = euler_gravity()
    npde = NumpyPDE(p)
    n_patches, P, H = 2, 15, 1
    S = P + 2 * H
    Q = euler_dg_state((n_patches, S, S, S), seed=9)
    dt, h = 1e-3, 0.05
    kern = exa.FVRusanovKernel(3, P, H, 5, 0, n_patches, pde=p.register(), mode=exa.FV_RUSANOV)
    got = np.ascontiguousarray(Q.copy())
    kern.time_step(got, dt, h)
    acc = np.zeros((n_patches, P, P, P, 5))
    core = (slice(None),) + (slice(H, H + P),) * 3
    for d in range(3):
        sh = lambda a, s_: np.roll(a, -s_, axis=1 + d)[core]
        qc, qp, qm = Q[core], sh(Q, 1), sh(Q, -1)
        lc, lp, lm = npde.maxeig(qc, d), npde.maxeig(qp, d), npde.maxeig(qm, d)
        Fc, Fp, Fm = npde.flux(qc, d), npde.flux(qp, d), npde.flux(qm, d)
        acc += 0.5 * (Fc + Fp) - 0.5 * np.maximum(lc, lp)[..., None] * (qp - qc)
        acc -= 0.5 * (Fm + Fc) - 0.5 * np.maximum(lm, lc)[..., None] * (qc - qm)
    want = Q.copy()
    want[core] = Q[core] - dt / h * acc + dt * npde.source(Q[core])
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-12


# ---- the exahype2::CellData flavour: out of place, PDE terms that see the volume centre and the time --------------------------------
class NumpyXtPDE:
    """SympyPDE with (x, t)-dependent terms, lambdified: flux(q, x, t, d), maxeig(q, x, t, d), source(q, x, t); x[..., 3]."""

    def __init__(self, spde):
        self.m = spde.n_vars
        args = list(spde.q) + list(spde.x) + [spde.t]
        self._f = [sympy.lambdify(args, spde.flux_exprs[d], "numpy") for d in range(spde.max_dim)]
        self._e = [sympy.lambdify(args, spde.eig_exprs[d], "numpy") for d in range(spde.max_dim)]
        self._s = sympy.lambdify(args, spde.source_exprs, "numpy") if spde.source_exprs is not None else None

    def _args(self, q, x, t):
        return [q[..., v] for v in range(self.m)] + [x[..., a] for a in range(3)] + [t]

    def flux(self, q, x, t, d):
        return np.stack([np.broadcast_to(o, q.shape[:-1]) for o in self._f[d](*self._args(q, x, t))], axis=-1)

    def maxeig(self, q, x, t, d):
        return np.broadcast_to(self._e[d](*self._args(q, x, t)), q.shape[:-1])

    def source(self, q, x, t):
        return np.stack([np.broadcast_to(o, q.shape[:-1]) for o in self._s(*self._args(q, x, t))], axis=-1)


def variable_coefficient_system(max_dim=2):
    """Two species advected with a velocity that depends on position and time, and a source that depends on both."""
    from exahype_amd.pde_codegen import SympyPDE
    vel = lambda x, t, d: 1 + sympy.Rational(1, 2) * x[d] + sympy.Rational(1, 4) * sympy.sin(t)
    return SympyPDE(2, flux=lambda q, x, t, d: [vel(x, t, d) * q[0], sympy.Rational(3, 4) * vel(x, t, d) * q[1]],
                    max_eigenvalue=lambda q, x, t, d: sympy.Abs(vel(x, t, d)),
                    source=lambda q, x, t: [x[0] * sympy.cos(t) - q[0] * x[1], q[0] - 2 * q[1] + t],
                    max_dim=max_dim, name="variable_coefficient")


def _volume_centres(centres, P, H, h, dim):
    """[n_patches][S]^dim[3]: exahype2::fv::getVolumeCentre for every volume of the array with halo (halo volumes continue the lattice)."""
    S = P + 2 * H
    idx = np.arange(S) - H + 0.5 - 0.5 * P
    x = np.zeros((len(centres),) + (S,) * dim + (3,))
    for a in range(dim):
        sh = [1] * (1 + dim)
        sh[1 + a] = S
        x[..., a] = centres[:, a].reshape((-1,) + (1,) * dim) + (idx * h).reshape(sh)
    return x


def test_position_and_time_dependent_terms_generate_an_xt_term_set():
    p = variable_coefficient_system()
    assert p.uses_xt and "HAS_XT = true" in p.source() and "flux_xt(const double* q, const double* x, double t" in p.source()
    assert not reaction_advection().uses_xt and "HAS_XT" not in reaction_advection().source()


@pytest.mark.gpu
def test_cell_data_flavour_out_of_place_equals_in_place():
    """exa_fv_time_step_device_oop with a built-in term set: QOut == the interior the in-place call leaves, bit for bit where both run
    the same kernel (faithful and corrected mode, the reference's configuration and a 3-D one with auxiliary variables), auxiliary variables
    copied, QIn untouched."""
    from exahype_amd import solvers as exa
    from tests.util import euler_patches, euler_ref2d_patches
    for dim, P, H, m, aux, pde, mode in ((2, 4, 1, 5, 5, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL), (2, 4, 1, 5, 5, exa.PDE_EULER_REF2D, exa.FV_RUSANOV),
                                         (3, 6, 1, 5, 2, exa.PDE_EULER, exa.FV_RUSANOV), (3, 15, 1, 5, 0, exa.PDE_EULER, exa.FV_RUSANOV)):
        n, S, V = 7, P + 2 * H, m + aux
        Q = euler_ref2d_patches(n, S, V, seed=3) if pde == exa.PDE_EULER_REF2D else euler_patches(n, dim, S, V, seed=4)
        k = exa.FVRusanovKernel(dim, P, H, m, aux, n, pde=pde, mode=mode)
        ref = np.ascontiguousarray(Q.copy())
        k.time_step(ref, 1e-3, 0.05)
        qin = np.ascontiguousarray(Q.copy())
        out = k.time_step_oop(qin, 1e-3, 0.05)
        core = (slice(None),) + (slice(H, H + P),) * dim
        assert out.shape == (n,) + (P,) * dim + (V,)
        if P == 15:     # the in-place call of this shape runs the plane-streaming kernel (cached 1/rho, p, c by fast reciprocal / square root)
            assert np.max(np.abs(out - ref[core])) / np.max(np.abs(ref)) < 1e-13
        else:
            assert np.array_equal(out, ref[core])
        assert np.array_equal(qin, Q) and np.array_equal(out[..., m:], Q[core][..., m:])


@pytest.mark.gpu
@pytest.mark.parametrize("dim,P", [(2, 5), (3, 4)])
def test_cell_data_flavour_position_and_time_dependent_terms_vs_numpy(dim, P):
    """Rusanov update with terms that depend on the volume centre and on t, against numpy with the SAME lambdified expressions
    (parity unpinned against the reference: its harness only declares these hooks, `Unit test/correctness_test.cpp:16-41`)."""
    from exahype_amd import solvers as exa
    p = variable_coefficient_system(max_dim=dim)
    npde = NumpyXtPDE(p)
    n, H = 3, 1
    S = P + 2 * H
    rng = np.random.default_rng(17)
    Q = 1.0 + 0.3 * rng.random((n,) + (S,) * dim + (2,))
    centres = rng.random((n, dim)) * 2 - 1
    dt, h, t = 2e-3, 0.07, 0.4
    kern = exa.FVRusanovKernel(dim, P, H, 2, 0, n, pde=p.register(), mode=exa.FV_RUSANOV)
    got = kern.time_step_oop(np.ascontiguousarray(Q), dt, h, t=t, centres=centres)
    x = _volume_centres(centres, P, H, h, dim)
    core = (slice(None),) + (slice(H, H + P),) * dim
    acc = np.zeros_like(Q[core])
    for d in range(dim):
        sh = lambda a, s_: np.roll(a, -s_, axis=1 + d)[core]
        qc, qp, qm = Q[core], sh(Q, 1), sh(Q, -1)
        xc, xp, xm = x[core], sh(x, 1), sh(x, -1)
        lc, lp, lm = npde.maxeig(qc, xc, t, d), npde.maxeig(qp, xp, t, d), npde.maxeig(qm, xm, t, d)
        Fc, Fp, Fm = npde.flux(qc, xc, t, d), npde.flux(qp, xp, t, d), npde.flux(qm, xm, t, d)
        acc += 0.5 * (Fc + Fp) - 0.5 * np.maximum(lc, lp)[..., None] * (qp - qc)
        acc -= 0.5 * (Fm + Fc) - 0.5 * np.maximum(lm, lc)[..., None] * (qc - qm)
    want = Q[core] - dt / h * acc + dt * npde.source(Q[core], x[core], t)
    assert np.max(np.abs(got - want)) < 1e-12
    # the in-place call of such a term set: patches centred at the origin, t = 0
    inplace = np.ascontiguousarray(Q.copy())
    kern.time_step(inplace, dt, h)
    assert np.array_equal(inplace[core], kern.time_step_oop(np.ascontiguousarray(Q), dt, h))


@pytest.mark.gpu
def test_fv_host_driver_hands_patch_centres_and_time_to_the_terms():
    """FVPatchGrid with a term set that sees position and time: the grid step carries the patch centres of the grid and the
    running time -- two steps == the out-of-place call on the array with filled halos"""
    import torch
    from exahype_amd import solvers as exa
    p = variable_coefficient_system(max_dim=2)
    grid, P, H = (3, 2), 4, 1
    fv = exa.FVPatchGrid(2, grid, P, H, 2, 0, p.register(), exa.FV_RUSANOV, length=1.5, origin=[0.2, -0.4], time=0.3)
    rng = np.random.default_rng(3)
    fv.set_interior(1.0 + 0.3 * rng.random(grid + (P, P, 2)))
    kern = exa.FVRusanovKernel(2, P, H, 2, 0, 6, pde=p.register(), mode=exa.FV_RUSANOV)
    h = 1.5 / (3 * P)
    assert abs(fv.h - h) < 1e-15
    idx = np.stack(np.meshgrid(np.arange(3), np.arange(2), indexing="ij"), axis=-1).reshape(-1, 2)
    centres = np.array([0.2, -0.4])[None, :] + (idx + 0.5) * P * h
    assert np.allclose(fv.centres.cpu().numpy(), centres, atol=1e-15)
    t = 0.3
    for dt in (1e-3, 2e-3):
        fv.fill_halos()
        want = kern.time_step_oop(fv.with_halo().reshape((6,) + fv.with_halo().shape[2:]), dt, h, t=t, centres=torch.as_tensor(centres, device="cuda"))
        fv.step(dt)
        t += dt
        assert np.max(np.abs(fv.interior().reshape(want.shape) - want.cpu().numpy())) < 1e-13
    assert abs(fv.time - t) < 1e-15
    # numpy in, with coordinates: staged through the device, the same update
    fv.fill_halos()
    Qh0 = fv.with_halo()
    qn = np.ascontiguousarray(Qh0.reshape((6,) + Qh0.shape[2:]).cpu().numpy())
    want = kern.time_step_oop(qn.copy(), 1e-3, h, t=t, centres=centres)
    kern.time_step(qn, 1e-3, h, t=t, centres=centres)
    assert np.max(np.abs(qn[:, H:H + P, H:H + P] - want)) < 1e-13
    # the CFL scan hands the volume centres and the time to the eigenvalue (exa_pde_eval_device_at)
    npde = NumpyXtPDE(p)
    xv = _volume_centres(centres, P, H, h, 2)
    Qh = fv.with_halo()
    qa = Qh.reshape((6,) + Qh.shape[2:]).cpu().numpy()
    inner = (slice(None), slice(H, H + P), slice(H, H + P))              # (the scan looks at the grid's own volumes, not at halo copies at shifted positions)
    want_lam = max(np.max(np.abs(npde.maxeig(qa, xv, fv.time, d)[inner])) for d in range(2))
    assert abs(fv.max_eigenvalue() - want_lam) < 1e-13                   # left behind by the step kernel (eigenvalues of the states it wrote, at t + dt)
    fv.invalidate()
    assert abs(fv.max_eigenvalue() - want_lam) < 1e-13                   # ... and by the scan pass
    assert exa._lib.load().exa_pde_flags(p.register()) == 1 and exa._lib.load().exa_pde_flags(exa.PDE_EULER) == 0
    # the terms really see the grid's coordinates: the same grid at the origin gives something else
    fv0 = exa.FVPatchGrid(2, grid, P, H, 2, 0, p.register(), exa.FV_RUSANOV, length=1.5)
    fv0.set_interior(1.0 + 0.3 * np.random.default_rng(3).random(grid + (P, P, 2)))
    fv0.step(1e-3)
    fv0.step(2e-3)
    assert np.max(np.abs(fv0.interior() - fv.interior())) > 1e-6


@pytest.mark.gpu
def test_cell_data_flavour_manufactured_source():
    """A state that is constant in space with S = (x0 cos t, 1 + x1): the Rusanov update is forward Euler in time, so after K steps
    q0 = 1 + x0 * sum_k dt cos(t_k), q1 = 2 + K dt (1 + x1) in every volume, to rounding -- position and time reach the term set."""
    from exahype_amd import solvers as exa
    from exahype_amd.pde_codegen import SympyPDE
    p = SympyPDE(2, flux=lambda q, x, t, d: [0 * q[0], 0 * q[1]], max_eigenvalue=lambda q, x, t, d: sympy.Integer(0),     # no flux, no dissipation
                 source=lambda q, x, t: [x[0] * sympy.cos(t), 1 + x[1]], max_dim=2, name="manufactured_source")
    n, P, H = 2, 6, 1
    S = P + 2 * H
    kern = exa.FVRusanovKernel(2, P, H, 2, 0, n, pde=p.register(), mode=exa.FV_RUSANOV)
    centres = np.array([[0.5, -1.0], [2.0, 0.25]])
    h, dt, K = 0.1, 0.05, 6
    Q = np.zeros((n, S, S, 2))
    Q[..., 0], Q[..., 1] = 1.0, 2.0
    x = _volume_centres(centres, P, H, h, 2)
    core = (slice(None), slice(H, H + P), slice(H, H + P))
    for k in range(K):
        out = kern.time_step_oop(np.ascontiguousarray(Q), dt, h, t=k * dt, centres=centres)
        Q[core] = out
        # (zero flux and zero wave speed: the halo does not enter)
    want0 = 1.0 + x[core][..., 0] * sum(dt * np.cos(k * dt) for k in range(K))
    want1 = 2.0 + K * dt * (1 + x[core][..., 1])
    assert np.max(np.abs(Q[core][..., 0] - want0)) < 1e-13 and np.max(np.abs(Q[core][..., 1] - want1)) < 1e-13


@pytest.mark.gpu
def test_cell_data_flavour_through_the_printer():
    """The recognised CellData statement list, dispatched: HIPPrinter(...).run_cell_data == FVRusanovKernel.time_step_oop."""
    from exahype_amd import KernelBuilder, solvers as exa
    from exahype_amd.printers import HIPPrinter
    from tests.ref_examples import cell_data_patch_update
    p = variable_coefficient_system()
    kb = cell_data_patch_update(KernelBuilder, dim=2, patch_size=5, halo_size=1, n_real=2, n_aux=0, n_patches=3)
    hp = HIPPrinter(kb, pde=p)
    rng = np.random.default_rng(23)
    Q = 1.0 + 0.3 * rng.random((3, 7, 7, 2))
    centres = rng.random((3, 2))
    out = hp.run_cell_data(np.ascontiguousarray(Q), 1e-3, t=0.3, cell_centre=centres, cell_size=0.5)
    kern = exa.FVRusanovKernel(2, 5, 1, 2, 0, 3, pde=p.register(), mode=exa.FV_RUSANOV)
    assert np.array_equal(out, kern.time_step_oop(np.ascontiguousarray(Q), 1e-3, 0.1, t=0.3, centres=centres))


# ---- non-conservative product ------------------------------------------------------------------------------------------------------------
def two_layer_like(max_dim=2):
    """A system with a flux AND a non-conservative product (the shape of two-layer shallow water: the coupling of the layers is B(q) grad q):
    q0_t + div(a q0) + k q1 grad q0 = 0,  q1_t + div(b q1) + k q0 grad q1 = 0."""
    from exahype_amd.pde_codegen import SympyPDE
    a, b, k = (1.0, 0.5, -0.25), (0.75, -0.5, 0.5), 0.3
    return SympyPDE(2, flux=lambda q, d: [a[d] * q[0], b[d] * q[1]], max_eigenvalue=lambda q, d: sympy.Float(1.5),
                    ncp=lambda q, dq, d: [k * q[1] * dq[0], k * q[0] * dq[1]], max_dim=max_dim, name="two_layer_like")


def test_ncp_reaches_the_generated_device_code():
    src = two_layer_like().source()
    assert "HAS_NCP = true" in src and "ncp(const double* q, const double* dq, int d, double* out)" in src and "*dq[0]*q[1];" in src
    assert "HAS_NCP" not in swe().source()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,P", [(2, 6), (3, 4)])
def test_ncp_fv_rusanov_vs_numpy(dim, P):
    """Corrected Rusanov with the path-conservative jump term D = B_d(mean) (q_R - q_L), half to either side of a face, against numpy
    with the same lambdified expressions (parity unpinned against the reference: its harness only names the ncp slot)."""
    from exahype_amd import solvers as exa
    p = two_layer_like(max_dim=dim)
    f_ncp = [sympy.lambdify(list(p.q) + list(p.dq), p.ncp_exprs[d], "numpy") for d in range(dim)]
    npde = NumpyPDE(p)
    n, H = 3, 1
    S = P + 2 * H
    Q = 1.0 + 0.3 * np.random.default_rng(41).random((n,) + (S,) * dim + (2,))
    dt, h = 2e-3, 0.05
    kern = exa.FVRusanovKernel(dim, P, H, 2, 0, n, pde=p.register(), mode=exa.FV_RUSANOV)
    got = np.ascontiguousarray(Q.copy())
    kern.time_step(got, dt, h)
    core = (slice(None),) + (slice(H, H + P),) * dim
    acc = np.zeros_like(Q[core])
    ncp = lambda qa, dq, d: np.stack([np.broadcast_to(o, qa.shape[:-1]) for o in f_ncp[d](qa[..., 0], qa[..., 1], dq[..., 0], dq[..., 1])], axis=-1)
    for d in range(dim):
        sh = lambda a, s_: np.roll(a, -s_, axis=1 + d)[core]
        qc, qp, qm = Q[core], sh(Q, 1), sh(Q, -1)
        lc, lp, lm = npde.maxeig(qc, d), npde.maxeig(qp, d), npde.maxeig(qm, d)
        Fc, Fp, Fm = npde.flux(qc, d), npde.flux(qp, d), npde.flux(qm, d)
        acc += 0.5 * (Fc + Fp) - 0.5 * np.maximum(lc, lp)[..., None] * (qp - qc)
        acc -= 0.5 * (Fm + Fc) - 0.5 * np.maximum(lm, lc)[..., None] * (qc - qm)
        acc += 0.5 * ncp(0.5 * (qc + qp), qp - qc, d) + 0.5 * ncp(0.5 * (qc + qm), qc - qm, d)
    want = Q.copy()
    want[core] = Q[core] - dt / h * acc
    assert np.max(np.abs(got - want)) < 1e-13
    # constant-coefficient advection written as a flux or as an ncp is the same scheme
    from exahype_amd.pde_codegen import SympyPDE
    a = (1.0, 0.5, -0.25)
    as_flux = SympyPDE(2, flux=lambda q, d: [a[d] * q[0], a[d] * q[1]], max_eigenvalue=lambda q, d: sympy.Float(1.0), max_dim=dim, name="adv_flux")
    as_ncp = SympyPDE(2, flux=lambda q, d: [0 * q[0], 0 * q[1]], max_eigenvalue=lambda q, d: sympy.Float(1.0),
                      ncp=lambda q, dq, d: [a[d] * dq[0], a[d] * dq[1]], max_dim=dim, name="adv_ncp")
    outs = []
    for pde in (as_flux, as_ncp):
        kk = exa.FVRusanovKernel(dim, P, H, 2, 0, n, pde=pde.register(), mode=exa.FV_RUSANOV)
        o = np.ascontiguousarray(Q.copy())
        kk.time_step(o, dt, h)
        outs.append(o)
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-14


# ---- ADER-DG with node coordinates, level times and a non-conservative product (exa_dg_plain.hpp, dg_stage_b_dense_kernel) ----------------
class OracleXtPDE:
    """A SympyPDE lambdified for oracle/aderdg_numpy.py step_xt: flux(q, x, t, a), maxeig, source, ncp with x a list of three arrays."""

    def __init__(self, spde):
        self.m = spde.n_vars
        xt = list(spde.x) + [spde.t]
        self._f = [sympy.lambdify(list(spde.q) + xt, spde.flux_exprs[d], "numpy") for d in range(spde.max_dim)]
        self._e = [sympy.lambdify(list(spde.q) + xt, spde.eig_exprs[d], "numpy") for d in range(spde.max_dim)]
        if spde.source_exprs is not None:
            s_ = sympy.lambdify(list(spde.q) + xt, spde.source_exprs, "numpy")
            self.source = lambda q, x, t: self._stack(s_(*self._qs(q), *x, t), q)
        if spde.ncp_exprs is not None:
            n_ = [sympy.lambdify(list(spde.q) + list(spde.dq) + xt, spde.ncp_exprs[d], "numpy") for d in range(spde.max_dim)]
            self.ncp = lambda q, dq, x, t, a: self._stack(n_[a](*self._qs(q), *self._qs(dq), *x, t), q)

    def _qs(self, q):
        return [q[..., v] for v in range(self.m)]

    @staticmethod
    def _stack(outs, q):
        return np.stack([np.broadcast_to(o, q.shape[:-1]) for o in outs], axis=-1)

    def flux(self, q, x, t, a):
        return self._stack(self._f[a](*self._qs(q), *x, t), q)

    def maxeig(self, q, x, t, a):
        return np.broadcast_to(self._e[a](*self._qs(q), *x, t), q.shape[:-1])


def coupled_xt_ncp_system(max_dim=3, with_ncp=True, with_xt=True):
    """Three coupled fields: advection with a velocity that depends on position and time, a non-conservative coupling B(q) grad q, a source that
    depends on both -- every slot of the harness's kernel (`Unit test/correctness_test.cpp:145-155`: Flux, ncp, Source, Eigen) at once."""
    from exahype_amd.pde_codegen import SympyPDE
    R = sympy.Rational
    if with_xt:
        vel = lambda x, t, d: (1, R(-1, 2), R(3, 4))[d] + R(3, 10) * sympy.sin(2 * x[(d + 1) % max_dim] + t)
        flux = lambda q, x, t, d: [vel(x, t, d) * q[0], R(3, 4) * vel(x, t, d) * q[1] + R(1, 5) * q[0] * q[2], R(1, 2) * vel(x, t, d) * q[2]]
        eig = lambda q, x, t, d: sympy.Abs(vel(x, t, d)) + R(1, 5) * sympy.Abs(q[0])
        source = lambda q, x, t: [x[0] * sympy.cos(t) - q[0] * x[1], q[0] - 2 * q[1] + t, R(1, 2) * q[1] * x[0]]
        ncp = (lambda q, dq, x, t, d: [R(3, 10) * q[1] * dq[0], (R(1, 5) + R(1, 10) * x[d]) * q[0] * dq[2], R(1, 4) * dq[1] * (1 + t)]) if with_ncp else None
    else:
        a = (1.0, -0.5, 0.75)
        flux = lambda q, d: [a[d] * q[0], 0.75 * a[d] * q[1] + 0.2 * q[0] * q[2], 0.5 * a[d] * q[2]]
        eig = lambda q, d: sympy.Float(abs(a[d])) + 0.2 * sympy.Abs(q[0])
        source = None
        ncp = lambda q, dq, d: [0.3 * q[1] * dq[0], 0.2 * q[0] * dq[2], 0.25 * dq[1]]
    return SympyPDE(3, flux=flux, max_eigenvalue=eig, source=source, ncp=ncp, max_dim=max_dim,
                    name="coupled_%s%s" % ("xt" if with_xt else "", "_ncp" if ncp is not None else ""))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,N,nc,with_ncp,with_xt", [(2, 3, (3, 2), False, True), (2, 4, (2, 3), True, True), (3, 3, (2, 2, 2), True, True),
                                                         (3, 6, (2, 1, 2), True, True), (3, 4, (1, 2, 2), True, False), (2, 8, (2, 2), True, True),
                                                         (3, 2, (2, 2, 2), True, True), (2, 2, (3, 2), True, True),      # N < dim + 1: qbar | Fbar_a need more LDS than N levels
                                                         (3, 8, (1, 2, 1), True, True), (3, 8, (2, 1, 1), False, True),   # cfg 4's order: the matrix-pipe kernel
                                                         (3, 6, (1, 1, 2), True, False)])
def test_aderdg_position_time_and_ncp_vs_numpy_oracle(dim, N, nc, with_ncp, with_xt):
    """ADER-DG for q_t + div F(q, x, t) + B(q, x, t) . grad q = S(q, x, t): node coordinates and level times reach the terms, the ncp enters the
    predictor, the time-averaged update and the Riemann solve.  Against oracle/aderdg_numpy.py step_xt with the SAME lambdified expressions
    (parity unpinned against the reference, which holds no ADER-DG); three steps with a non-zero origin and start time, full predictor and
    single-stage scheme."""
    from exahype_amd import solvers as exa
    from oracle import aderdg_numpy as A
    from oracle.dg_operators import operators
    p = coupled_xt_ncp_system(max_dim=dim, with_ncp=with_ncp, with_xt=with_xt)
    assert p.uses_xt == with_xt
    o = OracleXtPDE(p)
    rng = np.random.default_rng(100 * dim + N)
    u = 1.0 + 0.3 * rng.random(tuple(nc) + (N,) * dim + (3,))
    dx = [(0.9, 1.1, 0.7)[a] / nc[a] for a in range(dim)]
    origin = [0.25, -0.5, 1.0][:dim]
    dt = 0.03 * min(dx) / (2 * N - 1)
    for n_picard in ((-1, 0, 2) if not (dim == 3 and N > 6) else (-1, 2)):          # (3-D N = 8: the plain kernel of the single-stage scheme does not fit the LDS)
        s = exa.AderDgSolver(dim, N, nc, pde=p.register(), n_vars=3, dx=dx, n_picard=n_picard, origin=origin, time=0.4)
        # 3-D N = 6 / 8 with a Picard loop: the register-resident / the matrix-pipe kernel carries the coordinates and the ncp itself; the other
        # orders and the single-stage scheme take the plain kernel
        tuned = {6: "reg_kernel", 8: "m8_kernel"}.get(N) if (dim == 3 and n_picard != 0) else None
        assert (tuned or "plain") in s.stage_a_kernel_name()
        s.upload(u)
        ref, t = u.copy(), 0.4
        for k in range(3):
            s.step(dt * (1 + 0.1 * k))
            ref = A.step_xt(ref, dt * (1 + 0.1 * k), dx, operators(N), o, t=t, origin=origin, n_it=N if n_picard < 0 else n_picard)
            t += dt * (1 + 0.1 * k)
        assert abs(s.time - t) < 1e-15
        assert np.max(np.abs(s.download() - ref)) / np.max(np.abs(ref)) < 1e-10, n_picard
        # the CFL scan: the eigenvalue at the node coordinates and the current time
        x3 = A._coords(tuple(nc), N, operators(N), dx, origin)
        un = s.download()
        want_lam = max(np.max(np.abs(o.maxeig(un, x3, s.time, a) * np.ones(un.shape[:-1]))) for a in range(dim))
        assert abs(float(s.max_eigenvalue()[0]) - want_lam) < 1e-12
        assert s.lib.exa_pde_flags(p.register()) == (1 if with_xt else 0) + (2 if with_ncp else 0)
    # every slot is really in the kernels: the oracle without it is far from what they produced
    for slot in (["ncp"] if with_ncp else []) + (["source"] if with_xt else []):
        o2 = OracleXtPDE(p)
        delattr(o2, slot)
        ref0, t = u.copy(), 0.4
        for k in range(3):
            ref0 = A.step_xt(ref0, dt * (1 + 0.1 * k), dx, operators(N), o2, t=t, origin=origin, n_it=2)
            t += dt * (1 + 0.1 * k)
        assert np.max(np.abs(s.download() - ref0)) > 1e-7, slot


@pytest.mark.gpu
def test_aderdg_constant_coefficient_advection_as_flux_or_as_ncp():
    """a . grad q written as div(a q) or as the non-conservative product B = a: for a constant a and Rusanov's flux the two ADER-DG schemes
    coincide (the jump term of the path-conservative form is a (q+ - q-), the flux form's central part the same)."""
    from exahype_amd import solvers as exa
    from exahype_amd.pde_codegen import SympyPDE
    a = (1.0, 0.5, -0.25)
    as_flux = SympyPDE(2, flux=lambda q, d: [a[d] * q[0], a[d] * q[1]], max_eigenvalue=lambda q, d: sympy.Float(1.0), max_dim=3, name="dg_adv_flux")
    as_ncp = SympyPDE(2, flux=lambda q, d: [0 * q[0], 0 * q[1]], max_eigenvalue=lambda q, d: sympy.Float(1.0),
                      ncp=lambda q, dq, d: [a[d] * dq[0], a[d] * dq[1]], max_dim=3, name="dg_adv_ncp")
    N, nc = 4, (2, 2, 2)
    u = 1.0 + 0.3 * np.random.default_rng(8).random(nc + (N,) * 3 + (2,))
    outs = []
    for p in (as_flux, as_ncp):
        s = exa.AderDgSolver(3, N, nc, pde=p.register(), n_vars=2)
        s.upload(u)
        for _ in range(3):
            s.step(2e-3)
        outs.append(s.download())
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-12
    assert np.max(np.abs(outs[0] - u)) > 1e-3


@pytest.mark.gpu
def test_aderdg_manufactured_solution_with_position_and_time():
    """q(x, t) = 2 + sin(2 pi (x0 - t)) cos(2 pi x1) solves q_t + div(a q) = S(x, t) with a = (1, 1/2) and the source read off the exact solution;
    the DG solution converges to it at the scheme's order when coordinates and time reach the source (p = 3: error ratio ~16 per halving)."""
    from exahype_amd import solvers as exa
    from exahype_amd.pde_codegen import SympyPDE
    from oracle.dg_operators import operators
    pi = sympy.pi
    exact = lambda x, t: 2 + sympy.sin(2 * pi * (x[0] - t)) * sympy.cos(2 * pi * x[1])
    def src(q, x, t):
        e = exact(x, t)
        return [sympy.diff(e, t) + sympy.diff(e, x[0]) + sympy.Rational(1, 2) * sympy.diff(e, x[1])]
    p = SympyPDE(1, flux=lambda q, x, t, d: [(1, sympy.Rational(1, 2))[d] * q[0]], max_eigenvalue=lambda q, x, t, d: sympy.Float(1.0),
                 source=src, max_dim=2, name="manufactured_dg")
    N, errs = 4, []
    xi = operators(N)["xi"]
    f = sympy.lambdify(list(p.x[:2]) + [p.t], exact(p.x, p.t), "numpy")
    for n in (4, 8):
        X = (np.arange(n)[:, None] + xi[None, :]) / n
        x0, x1 = X[:, None, :, None], X[None, :, None, :]
        s = exa.AderDgSolver(2, N, (n, n), pde=p.register(), n_vars=1)
        s.upload(np.broadcast_to(f(x0, x1, 0.0), (n, n, N, N))[..., None].copy())
        steps = 8 * n // 4
        dt = 0.1 / steps
        for _ in range(steps):
            s.step(dt)
        errs.append(np.max(np.abs(s.download()[..., 0] - f(x0, x1, 0.1))))
    assert errs[1] < errs[0] / 8 and errs[1] < 2e-4, errs


@pytest.mark.gpu
def test_aderdg_position_dependent_terms_are_refused_where_the_plain_kernel_does_not_fit():
    """3-D, N = 7: two space-time images exceed LDS -- refused with the reason, not computed without the coordinates"""
    from exahype_amd import solvers as exa
    from exahype_amd._lib import ExaHypeHipError
    p = coupled_xt_ncp_system(max_dim=3)
    s = exa.AderDgSolver(3, 7, (1, 1, 1), pde=p.register(), n_vars=3)
    s.upload(1.0 + 0.1 * np.random.default_rng(0).random((1, 1, 1, 7, 7, 7, 3)))
    with pytest.raises(ExaHypeHipError, match="LDS"):
        s.step(1e-4)
    with pytest.raises(ValueError):
        exa.AderDgSolver(3, 6, (2, 2, 2), pde=p.register(), n_vars=3, one_kernel_step=True)



@pytest.mark.gpu
def test_limiter_hands_cell_centres_and_time_to_a_position_dependent_system():
    """SubcellLimiter with a term set whose terms depend on position / time: the masked FV patch update of the troubled cells
    gets the centre of each patch's DG cell and the step's start time (exa_fv_time_step_device_masked_at).  Against the same steps done by
    hand: DG step everywhere; troubled cells projected, updated by the UNMASKED patch kernel with explicit centres, reconstructed."""
    import ctypes as C
    import torch
    from exahype_amd import solvers as exa
    p = variable_coefficient_system(max_dim=2)
    pid = p.register()
    dim, N, nc = 2, 3, (4, 3)
    dx, origin, t0, dt = [0.3, 0.3], [0.2, -0.4], 0.3, 2e-3
    rng = np.random.default_rng(12)
    u0 = 1.0 + 0.3 * rng.random(nc + (N, N, 2))
    mask = np.zeros(nc, dtype=bool)
    mask[1, 2] = mask[3, 0] = True
    a = exa.AderDgSolver(dim, N, nc, pde=pid, n_vars=2, dx=dx, origin=origin, time=t0)
    lim = exa.SubcellLimiter(a, capacity=4)
    a.upload(u0)
    assert int(lim.step(dt, mask)) == 2
    got = lim.download()
    assert abs(a.time - (t0 + dt)) < 1e-15
    # by hand
    b = exa.AderDgSolver(dim, N, nc, pde=pid, n_vars=2, dx=dx, origin=origin, time=t0)
    b.upload(u0)
    limb = exa.SubcellLimiter(b, capacity=4)
    cells = torch.tensor([1 * 3 + 2, 3 * 3 + 0], dtype=torch.int64, device="cuda")
    patches = torch.zeros((2, limb.patch_doubles), dtype=torch.float64, device="cuda")
    exa.check(b.lib.exa_dg_project_patches(b._plan, C.c_void_p(b.u.data_ptr()), C.c_void_p(cells.data_ptr()), 2, C.c_void_p(patches.data_ptr()), None))
    b.step(dt)
    Ns = 2 * N - 1
    fv = exa.FVRusanovKernel(dim, Ns, 1, 2, 0, 2, pde=pid, mode=exa.FV_RUSANOV)
    centres = torch.tensor([[origin[0] + 1.5 * dx[0], origin[1] + 2.5 * dx[1]], [origin[0] + 3.5 * dx[0], origin[1] + 0.5 * dx[1]]], dtype=torch.float64, device="cuda")
    fv.time_step(patches.reshape(-1), dt, dx[0] / Ns, t=t0, centres=centres)
    exa.check(b.lib.exa_dg_reconstruct_patches(b._plan, C.c_void_p(patches.data_ptr()), C.c_void_p(cells.data_ptr()), 2, C.c_void_p(b.u.data_ptr()), None))
    want = b.download()
    assert np.max(np.abs(got - want)) < 1e-13
    c = exa.AderDgSolver(dim, N, nc, pde=pid, n_vars=2, dx=dx, origin=origin, time=t0)
    c.upload(u0)
    c.step(dt)
    assert np.max(np.abs(got[1, 2] - c.download()[1, 2])) > 1e-6         # (a troubled cell did take the FV result)
    assert np.array_equal(got[0, 0], c.download()[0, 0])                   # (an untroubled one the DG result)
