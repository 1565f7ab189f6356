"""FV Rusanov patch update through the reference's operator surface on an MI355X.

The statement list below is the one of the reference's example (`examples/Batched_stateless.py:9-35`: copy-in, flux, eigenvalue, flux difference,
Rusanov dissipation, copy-out) under other variable names -- it is what a user of the reference writes (`KernelBuilder` + opaque PDE-term functions) and
what the recogniser of `HIPPrinter` has to match; instead of printing C++
(`CPPPrinter`) or MLIR, `HIPPrinter` recognises the scheme and dispatches it to the fused HIP kernel.  Q keeps the reference's layout
[patch][(P + 2H)^dim][n_real + n_aux] (halo included, variable fastest) and is updated in place, interior only.

usage: python examples/fv_rusanov_patches.py [n_patches = 4096]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sympy.codegen.ast import integer, none, real

from exahype import KernelBuilder
from exahype.printers import HIPPrinter


def main(n_patches=4096, dim=2, P=4, H=1, n_real=5, n_aux=5):
    kb = KernelBuilder(dim, P, H, n_real, n_aux, n_patches=n_patches)
    U, W = kb.item("U"), kb.item("W")                       # state with halo, working copy
    flux = kb.directional_item("flux")                      # per direction: one flux vector per volume
    speed = kb.directional_item("speed", struct=False)      # ... and one wave speed
    tau = kb.const("tau")
    axis = kb.directional_const("axis", list(range(dim)))
    F = kb.function("Flux", parameter_types=[U, real, U], return_type=integer)
    lam = kb.function("maxEigenvalue", parameter_types=[U, real], return_type=real)
    mx = kb.function("max", parameter_types=[U, U], return_type=none)
    kb.single(W[0], U[0])
    kb.directional(F(W[0], axis, flux[0]))
    kb.directional(speed[0], lam(W[0], axis))
    kb.directional(W[0], W[0] + 0.5 * (flux[-1] - flux[1]))
    kb.directional(W[0], W[0] + 0.5 * tau * (-mx(speed[-1], speed[0]) * (U[0] - U[-1]) + mx(speed[1], speed[0]) * (U[0] - U[1])), struct=True)
    kb.single(U[0], W[0])

    S = P + 2 * H
    rng = np.random.default_rng(1)
    Q = np.empty((n_patches,) + (S,) * dim + (n_real + n_aux,))
    Q[..., 0] = 1.0 + 0.1 * rng.random(Q.shape[:-1])        # density
    Q[..., 1:n_real - 1] = 0.05 * rng.standard_normal(Q.shape[:-1] + (n_real - 2,))
    Q[..., n_real - 1] = 2.5 + 0.1 * rng.random(Q.shape[:-1])
    Q[..., n_real:] = 0.0
    before = Q.copy()
    printer = HIPPrinter(kb, "time_step")                   # same constructor as CPPPrinter(kernel, function_name)
    print("recognised scheme:", printer.scheme)
    printer.run(Q, 1e-3)                                    # == time_step(Q, dt) of the generated C++, on the GPU
    inner = (slice(None),) + (slice(H, H + P),) * dim
    halo_untouched = np.array_equal(np.delete(Q, np.s_[H:H + P], axis=1), np.delete(before, np.s_[H:H + P], axis=1))
    print("interior changed by at most %.3e, halo layers untouched along axis 0: %s" % (np.abs(Q[inner] - before[inner]).max(), halo_untouched))
    return Q


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
