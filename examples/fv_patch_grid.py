"""A periodic Cartesian grid of FV patches advanced to t_end under the CFL condition -- the role of the enclave task around the generated
`time_step`: halo of every patch from its neighbours, patch kernel, eigenvalue reduction for the next dt.

`FVPatchGrid` keeps the states halo-less in HBM; a step is ONE launch (the patch kernel takes the halo states from the neighbours' interior
layers on chip and reduces the eigenvalues of the states it writes), and the time loop reads one double per step.

usage: python examples/fv_patch_grid.py [patches per axis = 32] [t_end = 0.02]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from exahype_amd import solvers as exa


def main(g=32, t_end=0.02, dim=2, P=4):
    grid = exa.FVPatchGrid(dim, (g,) * dim, P, halo_size=1, n_real=5, n_aux=0, pde=exa.PDE_EULER, mode=exa.FV_RUSANOV, length=1.0)
    # volume centres of the whole grid: [g.., P.., dim]
    ax = (np.arange(g * P) + 0.5) * grid.h
    X = np.stack(np.meshgrid(*([ax] * dim), indexing="ij"), axis=-1)
    rho = 1.0 + 0.2 * np.sin(2 * np.pi * X.sum(-1))
    q = np.zeros(X.shape[:-1] + (5,))
    q[..., 0] = rho
    q[..., 1:1 + dim] = rho[..., None]                      # velocity (1, 1)
    q[..., 4] = 1.0 / 0.4 + 0.5 * dim * rho
    # [G.., V] on the volume grid -> [g.., P.., V] patch-major
    shp = sum(((g, P) for _ in range(dim)), ())
    perm = tuple(range(0, 2 * dim, 2)) + tuple(range(1, 2 * dim, 2)) + (2 * dim,)
    grid.set_interior(q.reshape(shp + (5,)).transpose(perm))
    m0 = grid.interior().sum(axis=tuple(range(2 * dim))) * grid.h ** dim
    steps = grid.run(t_end, cfl=0.4)
    m1 = grid.interior().sum(axis=tuple(range(2 * dim))) * grid.h ** dim
    print("%d steps to t = %.4f on %d patches; mass of (rho, rho u, rho v, rho w, E) before / after:" % (steps, grid.time, g ** dim))
    print("  ", m0, "\n  ", m1)
    return m0, m1


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32, float(sys.argv[2]) if len(sys.argv) > 2 else 0.02)
