"""A flux / eigenvalue system given as SymPy expressions, advanced by the ADER-DG kernels (predictor Picard loop, volume integral, face Riemann
solve, corrector) on an MI355X.

`SympyPDE` turns the expressions into a device term set (common sub-expressions across the directions cached per node, fast reciprocals, the
per-lane-normal flux as straight-line code), compiles it with hipcc for gfx950 and registers it; `AderDgSolver` then runs the same kernels the
built-in Euler term set uses.  On the benchmark configuration the generated set reaches 0.96x the hand-written one (bench.py --config cfg2_sympy).

usage: python examples/aderdg_sympy_euler.py [cells per axis = 8] [order p = 5] [steps = 5]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sympy

from exahype_amd import solvers as exa
from exahype_amd.pde_codegen import SympyPDE

GAMMA = sympy.Float(1.4)


def pressure(q):
    return (GAMMA - 1) * (q[4] - (q[1] ** 2 + q[2] ** 2 + q[3] ** 2) / (2 * q[0]))


def flux(q, d):                                   # q = (rho, rho u, rho v, rho w, E); d = 0-based normal
    un, p = q[d + 1] / q[0], pressure(q)
    f = [un * q[0], un * q[1], un * q[2], un * q[3], un * (q[4] + p)]
    f[d + 1] += p
    return f


def max_eigenvalue(q, d):
    return sympy.Abs(q[d + 1] / q[0]) + sympy.sqrt(GAMMA * pressure(q) / q[0])


def main(cells=8, p=5, steps=5):
    import torch
    N = p + 1
    pde = SympyPDE(5, flux, max_eigenvalue, max_dim=3, name="euler_from_sympy")
    solver = exa.AderDgSolver(3, N, (cells,) * 3, pde=pde.register(), n_vars=5)     # periodic unit cube
    # smooth density wave moving with (1, 1, 1): rho = 1 + 0.2 sin(2 pi (x + y + z)), p = 1
    X = solver.node_positions().reshape(solver.u.shape[:-1] + (3,))
    rho = 1.0 + 0.2 * torch.sin(2 * np.pi * X.sum(-1))
    solver.u[..., 0] = rho
    for a in range(3):
        solver.u[..., 1 + a] = rho
    solver.u[..., 4] = 1.0 / 0.4 + 1.5 * rho
    w = torch.as_tensor(solver.operators()["w"], device=solver.u.device)

    def mass():                                   # integral of every variable over the cube (quadrature weights per axis)
        u = solver.u
        for _ in range(3):
            u = torch.tensordot(u, w, dims=([3], [0]))          # nodes of one axis at a time: [cx, cy, cz, (nodes..), v]
        return u.sum((0, 1, 2)).cpu().numpy() / cells ** 3

    m0 = mass()
    n = solver.run(t_end=steps * 0.2 * solver.dx[0] / ((2 * N - 1) * 3 * 3.0), cfl=0.2)
    m1 = mass()
    print("kernel:", solver.stage_a_kernel_name())
    print("%d steps to t = %.4e; conservation of (rho, rho u, rho v, rho w, E): max relative drift %.2e"
          % (n, solver.time, float(np.max(np.abs(m1 - m0) / np.abs(m0)))))
    return m0, m1


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)
