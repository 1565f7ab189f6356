"""Shared input generators for the parity tests (seeded, admissible Euler states)."""
import numpy as np


def euler_dg_state(shape, seed, amp=0.2):
    """u[..., 5]: smooth-ish random admissible state (rho, m0, m1, m2, E)."""
    rng = np.random.default_rng(seed)
    u = np.zeros(tuple(shape) + (5,))
    rho = 1.0 + amp * rng.random(shape)
    u[..., 0] = rho
    vel = [0.4 * rng.random(shape) - 0.2 for _ in range(3)]
    for a in range(3):
        u[..., 1 + a] = rho * vel[a]
    p = 1.0 + amp * rng.random(shape)
    u[..., 4] = p / 0.4 + 0.5 * rho * sum(v * v for v in vel)
    return u


def euler_ref2d_patches(n_patches, S, V, seed):
    """Q[n_patches, S, S, V] with (rho, rho u, rho v, E) in vars 0..3, noise elsewhere."""
    rng = np.random.default_rng(seed)
    sh = (n_patches, S, S)
    Q = rng.uniform(-1, 1, sh + (V,))
    rho = rng.uniform(0.5, 2.0, sh); u = rng.uniform(-1, 1, sh); v = rng.uniform(-1, 1, sh); p = rng.uniform(0.5, 2.0, sh)
    Q[..., 0] = rho; Q[..., 1] = rho * u; Q[..., 2] = rho * v; Q[..., 3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    return Q


def euler_patches(n_patches, dim, S, V, seed):
    rng = np.random.default_rng(seed)
    sh = (n_patches,) + (S,) * dim
    Q = rng.uniform(-1, 1, sh + (V,))
    Q[..., :5] = euler_dg_state(sh, seed + 1, amp=0.5)
    return Q


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
