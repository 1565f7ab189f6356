"""Statement-by-statement lowering of builder states that are not a recognised scheme (exahype_amd/printers/lowering.py; the
reference lowers EVERY builder state, `exahype/printers/CPPPrinter.py:84-137`).

What pins what: the loop ranges, AoS strides and expression text come from this package's CPPPrinter, which is byte-identical
to the reference's output (tests/golden/cppprinter_*.txt, tests/test_operator_surface.py); the GPU execution of that text is
compared BIT FOR BIT with oracle/statement_eval.py, which evaluates the same text with numpy in the same operation order and is
itself pinned below against a stencil written by hand.  The reference's generated C++ cannot serve as the executable oracle:
its `time_step` allocates the primary array inside the function and takes the first constant under the array's type
(SURVEY.md Appendix B-7), so it computes on uninitialised memory."""
import re

import numpy as np
import pytest

from exahype_amd import KernelBuilder
from exahype_amd.printers import CPPPrinter, HIPPrinter, UnrecognisedKernel
from exahype_amd.printers.lowering import LoweringRefused, StatementLowering
from oracle.statement_eval import run_statements


def central_flux_update(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=1, n_patches=3):
    """A non-Rusanov conservative update: F_d = c q^2 + q/4 over the full range along d, then q -= (F_d[+1] - F_d[-1]) / 2."""
    k = KernelBuilder(dim=dim, patch_size=patch_size, halo_size=halo_size, n_real=n_real, n_aux=n_aux, n_patches=n_patches)
    Q = k.item('Q')
    F = k.directional_item('F')
    c = k.const('c')
    k.directional(F[0], c * Q[0] * Q[0] + 0.25 * Q[0])
    k.directional(Q[0], Q[0] - 0.5 * (F[1] - F[-1]))
    return k


def weighted_smoothing(dim=3, patch_size=3, halo_size=2, n_real=1, n_aux=2, n_patches=2):
    """Scalar temporaries (struct=False), two input constants, a directional constant inside the arithmetic, halo 2."""
    k = KernelBuilder(dim=dim, patch_size=patch_size, halo_size=halo_size, n_real=n_real, n_aux=n_aux, n_patches=n_patches)
    Q = k.item('Q')
    W = k.directional_item('W', struct=False)
    a = k.const('a')
    b = k.const('b')
    axis = k.directional_const('axis', list(range(dim)))
    k.directional(W[0], a * Q[0] + b * (1 + axis))
    k.directional(Q[0], Q[0] + (W[1] - 2 * W[0] + W[-1]) / (2 + axis), struct=True)
    return k


def _oracle(k, q, consts):
    L = StatementLowering(k)
    arrs = {a: (q if a == L.primary else np.zeros(L.arrays[a][0])) for a in L.array_order()}
    lits = {m.group(1): int(m.group(2)) for m in (re.match(r"int (\w+) = (\d+);", lit) for lit in k.literals) if m}
    run_statements(L.statements, [str(i) for i in k.indexes], arrs, dict(zip(L.consts, consts)), lits, functions=L.functions)
    return q


def _euler_flux(q, d):
    irho = 1 / q[0]
    p = 0.4 * (q[3] - 0.5 * irho * (q[1] ** 2 + q[2] ** 2))
    c = irho * q[d + 1]
    f = [c * q[0], c * q[1], c * q[2], c * q[3] + c * p]
    f[d + 1] = f[d + 1] + p
    return f


def _euler_eig(q, d):
    import sympy
    irho = 1 / q[0]
    p = 0.4 * (q[3] - 0.5 * irho * (q[1] ** 2 + q[2] ** 2))
    return sympy.Abs(q[d + 1] * irho) + sympy.sqrt(1.4 * p * irho)


def rusanov_with_bodies_and_source(n_patches=3, with_bodies=True):
    """The statement list of the reference's example (`examples/Batched_stateless.py:25-35`: copy, Flux / maxEigenvalue per direction, central
    flux difference, `max`-weighted dissipation, copy back) with the PDE terms given as SymPy BODIES, plus one statement pair the recognised
    Rusanov template does not know: an algebraic source `Source(Qc, S); Qc += dt * S`.  (The working copy is the THIRD item: the reference
    addresses items[1] as the halo-less array with `patch - 1`, Appendix B-6.)"""
    b = dict(body=_euler_flux), dict(body=_euler_eig), dict(body=lambda q: [0 * q[0], -0.3 * q[1], -0.3 * q[2], -0.1 * q[3] * q[0]])
    if not with_bodies:
        b = {}, {}, {}
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=4, n_aux=1, n_patches=n_patches)
    Q = k.item('Q')
    k.item('Qout')
    Qc = k.item('Qc')
    F = k.directional_item('F')
    L = k.directional_item('L', struct=False)
    S = k.directional_item('S')
    dt = k.const('dt')
    normal = k.directional_const('normal', [0, 1])
    Flux = k.function('Flux', **b[0])
    Eig = k.function('maxEigenvalue', **b[1])
    Max = k.function('max')
    Src = k.function('Source', **b[2])
    k.single(Qc[0], Q[0])                          # (interior only at the reference's HEAD: the terms below read Q, whose halo the caller filled)
    k.directional(Flux(Q[0], normal, F[0]))
    k.directional(L[0], Eig(Q[0], normal))
    k.directional(Qc[0], Qc[0] + 0.5 * (F[-1] - F[1]))
    left = -Max(L[-1], L[0]) * (Q[0] - Q[-1])
    right = -Max(L[1], L[0]) * (Q[0] - Q[1])
    k.directional(Qc[0], Qc[0] + 0.5 * dt * (left - right), struct=True)
    k.single(Src(Qc[0], S[0]), direction=1)
    k.single(Qc[0], Qc[0] + dt * S[0], direction=1, struct=True)
    k.single(Q[0], Qc[0])
    return k


def test_evaluator_equals_a_hand_written_stencil():
    k = central_flux_update()
    rng = np.random.default_rng(0)
    q0 = rng.random(3 * 6 * 6 * 3)
    got = _oracle(k, q0.copy(), [0.3])
    A = q0.reshape(3, 6, 6, 3)
    f = 0.3 * A[..., :2] * A[..., :2] + 0.25 * A[..., :2]
    B = A.copy()
    B[:, 1:5, 1:5, :2] = (-0.5 * f[:, 2:6, 1:5] + 0.5 * f[:, 0:4, 1:5]) + A[:, 1:5, 1:5, :2]
    B2 = B.copy()
    B2[:, 1:5, 1:5, :2] = (-0.5 * f[:, 1:5, 2:6] + 0.5 * f[:, 1:5, 0:4]) + B[:, 1:5, 1:5, :2]
    assert np.array_equal(got, B2.reshape(-1))
    assert np.array_equal(got.reshape(3, 6, 6, 3)[..., 2], A[..., 2])          # the auxiliary variable is not touched
    assert np.array_equal(got.reshape(3, 6, 6, 3)[:, 0], A[:, 0])              # nor the halo


def test_generated_kernels_carry_the_reference_text_and_ranges():
    """Every statement line of the reference-compatible C++ text appears verbatim in a generated kernel (powers aside: the
    reference prints Python's `**`), and the kernel's index decoding spans the loop nest's ranges."""
    k = weighted_smoothing()
    p = HIPPrinter(k)
    assert p.scheme == "statements" and "UnrecognisedKernel" not in p.code
    ref = CPPPrinter(k).code
    stmts = [ln.strip() for ln in ref.splitlines() if re.match(r"\s+\w+\[.*\] = .*;", ln)]
    assert len(stmts) == 2 * k.dim
    for ln in stmts:
        assert ln in p.code, ln
    loops = re.findall(r"for \(int (\w+) = (\d+); \w+ < (\d+);", ref)
    kern = re.findall(r"const int (\w+) = (\d+) \+ \(int\)\(r_ % (\d+)\)", p.code)
    assert sorted((n, int(lo), int(hi) - int(lo)) for n, lo, hi in loops) == sorted((n, int(lo), int(e)) for n, lo, e in kern)
    assert "const double axis = 0;" in p.code and "const double axis = 2;" in p.code      # the directional constant, per statement


def test_what_cannot_run_in_parallel_or_in_bounds_is_refused():
    # the reference's own example: opaque PDE functions (it goes through the recognised Rusanov scheme instead)
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q, T = k.item('Q'), k.directional_item('T')
    f = k.function('Flux')
    k.directional(T[0], f(Q[0]))
    with pytest.raises(LoweringRefused, match="opaque function `Flux`"):
        StatementLowering(k)
    with pytest.raises(UnrecognisedKernel, match="opaque function"):
        HIPPrinter(k)
    with pytest.raises(LoweringRefused, match="opaque function `Flux`.*SymPy body"):
        StatementLowering(rusanov_with_bodies_and_source(with_bodies=False))
    # a call that reads, by address, the array the statement writes: other lanes write those variables
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q = k.item('Q')
    g = k.function('G', body=lambda q: q[0] + q[1])
    k.single(Q[0], Q[0] + g(Q[0]))
    with pytest.raises(LoweringRefused, match="loop-carried dependence"):
        StatementLowering(k)
    # a list-valued body needs its out-parameter
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q, T = k.item('Q'), k.directional_item('T')
    f = k.function('Flux', body=lambda q: [q[0], q[1]])
    k.directional(T[0], f(Q[0]))
    with pytest.raises(LoweringRefused, match="returns a list"):
        StatementLowering(k)
    # a body that returns more entries than its out-parameter holds per volume would write into the next volume
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q, T = k.item('Q'), k.directional_item('T')
    f = k.function('Flux', body=lambda q, n: [q[0], q[1], q[0] * q[1]])
    ax = k.directional_const('ax', [0, 1])
    k.directional(f(Q[0], ax, T[0]))
    with pytest.raises(LoweringRefused, match="returns 3 expressions.*holds 2 entries"):
        StatementLowering(k)
    # a second item is the reference's halo-less array, addressed with `patch - 1`: out of bounds for the first patch (B-6)
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q, C = k.item('Q'), k.item('Qcopy')
    k.single(C[0], Q[0])
    with pytest.raises(LoweringRefused, match="out of bounds"):
        StatementLowering(k)
    # a loop-carried dependence: the sequential nest of the reference propagates values along i, a parallel launch does not
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q = k.item('Q')
    k.directional(Q[0], Q[-1])
    with pytest.raises(LoweringRefused, match="loop-carried dependence"):
        StatementLowering(k)
    # members of a host object
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    D = k.item('Data')
    Q = k.item('Q', parent=D)
    k.single(Q[0], 2 * Q[0])
    with pytest.raises(LoweringRefused, match="parent object"):
        StatementLowering(k)
    # an offset that leaves the halo
    k = KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=2, n_aux=0)
    Q, T = k.item('Q'), k.directional_item('T')
    k.directional(T[0], Q[2] - Q[0])
    with pytest.raises(LoweringRefused, match="out of bounds"):
        StatementLowering(k)


def test_lowering_needs_a_gpu_to_run():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less machines")
    with pytest.raises(RuntimeError, match="GPU"):
        HIPPrinter(central_flux_update()).run(np.zeros(3 * 6 * 6 * 3), 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("make,consts", [(central_flux_update, (0.3,)), (weighted_smoothing, (0.7, -0.2)),
                                          (lambda: central_flux_update(dim=3, patch_size=5, halo_size=1, n_real=3, n_aux=0, n_patches=4), (1.1,)),
                                          (lambda: weighted_smoothing(dim=2, patch_size=8, halo_size=1, n_real=2, n_aux=0, n_patches=1), (0.5, 0.25))])
def test_lowered_statements_equal_the_evaluator_bit_for_bit(make, consts):
    import torch
    k = make()
    p = HIPPrinter(k)
    assert p.scheme == "statements"
    n = StatementLowering(k).arrays[k.items[0]][0]
    q0 = np.random.default_rng(5).random(n) + 0.5
    want = _oracle(k, q0.copy(), list(consts))
    got = q0.copy()
    p.run(got, *consts)                                            # numpy array, staged
    assert np.array_equal(got, want)
    dev = torch.as_tensor(q0, device="cuda")
    p.run(dev, *consts, steps=2)                                   # device-resident, two calls
    want2 = _oracle(k, want.copy(), list(consts))
    assert np.array_equal(dev.cpu().numpy(), want2)
    with pytest.raises(TypeError):
        p.run(got)                                                 # constants missing


def test_statement_list_with_function_bodies_is_lowered_with_the_reference_calling_convention():
    """Functions with SymPy bodies become `__device__` functions, the statements keep the reference's text: array arguments by address
    (ONE `&`: `Unit test/test.cpp:25,45`), the directional constant as the normal, a bare call with its out-parameter."""
    import sympy
    k = rusanov_with_bodies_and_source()
    hp = HIPPrinter(k)                        # one statement more than the recognised Rusanov template knows: the printer lowers it
    assert hp.scheme == "statements" and "Source(&Qc[" in hp.code
    L = StatementLowering(k)
    src = L.source()
    assert "Flux(&Q[180*patch + 30*i + 5*j], normal, &F_x[144*patch + 24*i + 4*j]);" in src
    assert "L_y[36*patch + 6*i + 1*j] = maxEigenvalue(&Q[180*patch + 30*i + 5*j], normal);" in src
    assert "Source(&Qc[180*patch + 30*i + 5*j], &S_x[144*patch + 24*i + 4*j]);" in src
    assert "max(&L_x[36*patch + 6*(i - 1) + 1*j], &L_x[36*patch + 6*i + 1*j])" in src and "&&" not in src
    assert "static __device__ inline void Flux(double* p0, int n1, double* o2)" in src
    assert "static __device__ inline double maxEigenvalue(double* p0, int n1)" in src
    assert "static __device__ inline double max(double* p0, double* p1)" in src          # Functions.cpp:64-66, no body needed
    assert set(L.functions) == {"Flux", "maxEigenvalue", "max", "Source"} and L.functions["Flux"]["widths"] == [4, 0, 4]
    # the generated function bodies ARE the user's expressions (evaluated through the evaluator's own reading of the C text)
    from oracle.statement_eval import device_functions
    fn = device_functions(L.functions)
    rng = np.random.default_rng(2)
    qv = np.array([1.0 + rng.random(), rng.random() - 0.5, rng.random() - 0.5, 3.0 + rng.random()])
    qs = sympy.symbols("q0:4")
    for d in range(2):
        out = np.zeros(4)
        fn["Flux"](qv, d, out)
        want = [float(e.subs(dict(zip(qs, qv)))) for e in _euler_flux(qs, d)]
        assert np.allclose(out, want, rtol=1e-14, atol=1e-15)
        assert abs(fn["maxEigenvalue"](qv, d) - float(_euler_eig(qs, d).subs(dict(zip(qs, qv))))) < 1e-14
    assert fn["max"](np.array([0.25]), np.array([0.75])) == 0.75


@pytest.mark.gpu
def test_lowered_function_calls_equal_the_evaluator_bit_for_bit():
    """`examples/Batched_stateless.py:25-35` + a source statement, functions inlined from their SymPy bodies, lowered
    statement by statement and run on the GPU: bit-equal to oracle/statement_eval.py evaluating the same text (the device functions' text
    included), two steps on a device-resident array."""
    import torch
    k = rusanov_with_bodies_and_source()
    L = StatementLowering(k)
    n = L.arrays["Q"][0]
    rng = np.random.default_rng(9)
    q0 = np.zeros((3, 6, 6, 5))
    q0[..., 0] = 1.0 + 0.3 * rng.random(q0.shape[:-1])
    q0[..., 1] = 0.3 * rng.random(q0.shape[:-1]) - 0.15
    q0[..., 2] = 0.3 * rng.random(q0.shape[:-1]) - 0.15
    q0[..., 3] = 2.5 + 0.5 * rng.random(q0.shape[:-1])
    q0[..., 4] = rng.random(q0.shape[:-1])
    q0 = q0.reshape(-1)
    assert q0.size == n
    want = _oracle(k, q0.copy(), [1e-2])
    assert np.isfinite(want).all() and not np.array_equal(want, q0)
    L.bind()
    got = q0.copy()
    L.run(got, 1e-2)
    assert np.array_equal(got, want)
    dev = torch.as_tensor(q0, device="cuda")
    L.run(dev, 1e-2)
    L.run(dev, 1e-2)
    assert np.array_equal(dev.cpu().numpy(), _oracle(k, want.copy(), [1e-2]))
    # the source statement is in there: without it the result differs
    k0 = rusanov_with_bodies_and_source()
    k0.function_bodies["Source"] = lambda q: [0 * q[0], 0 * q[1], 0 * q[2], 0 * q[3]]
    assert not np.array_equal(_oracle(k0, q0.copy(), [1e-2]), want)
