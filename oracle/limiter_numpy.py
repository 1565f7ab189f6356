"""ORACLE (test infrastructure only -- never imported by the product path).

FV subcell limiter glue of BASELINE configs[4]: projection of a DG cell onto
N_s = 2p+1 equal subcells per axis, constrained least-squares reconstruction, and the limited step
(troubled cells take the FV Rusanov patch update of their projected data instead of the DG result).

Reference anchor: none -- /root/reference holds no limiter; "parity unpinned".  The FV
patch update used inside IS the reference's kernel shape (`Unit test/test.cpp`, corrected form).
Pinned by the identities in tests/test_limiter.py (R P = I on degree <= p data, mean preservation).
"""
import numpy as np

from .dg_operators import gauss_legendre_01, lagrange_eval


def projection_matrix(xi, Ns):
    """P[s][i] = N_s * int_{s/N_s}^{(s+1)/N_s} phi_i  (Gauss-Legendre on every subinterval: exact)."""
    N = len(xi)
    g, gw = gauss_legendre_01(N)
    P = np.zeros((Ns, N))
    for s in range(Ns):
        for k in range(N):
            x = (s + g[k]) / Ns
            P[s] += gw[k] * lagrange_eval(xi, x)
    return P


def reconstruction_matrix(P, w):
    """R = argmin ||P u - v||^2  s.t.  w.u = mean(v)   (KKT system), as a matrix N x N_s."""
    Ns, N = P.shape
    K = np.zeros((N + 1, N + 1))
    K[:N, :N] = 2 * P.T @ P
    K[:N, N] = w
    K[N, :N] = w
    rhs = np.zeros((N + 1, Ns))
    rhs[:N] = 2 * P.T
    rhs[N] = 1.0 / Ns
    return np.linalg.solve(K, rhs)[:N]


def apply_all_axes(M, a, dim, first_axis):
    for d in range(dim):
        a = np.moveaxis(np.tensordot(M, a, axes=([1], [first_axis + d])), 0, first_axis + d)
    return a


def limited_step(u, mask, dt, dx, ops, pde, fv_update, n_it=None):
    """One step of the limited scheme on a periodic grid.
    u[grid.., nodes.., m]; mask[grid..] bool (troubled); fv_update(patch[S..,m], dt, h) -> patch (in place semantics).
    Untroubled cells: ADER-DG step.  Troubled cells: project u^n (own + face neighbours' boundary layers) onto
    subcells, one FV Rusanov update with h = dx/N_s, reconstruct."""
    from . import aderdg_numpy as A
    dim = (u.ndim - 1) // 2
    N = ops["N"]
    Ns = 2 * (N - 1) + 1
    P = projection_matrix(ops["xi"], Ns)
    R = reconstruction_matrix(P, ops["w"])
    unew = A.step(u, dt, dx, ops, pde, n_it)
    proj = apply_all_axes(P, u, dim, dim)                                  # [grid.., Ns.., m]
    S = Ns + 2
    if max(dx[:dim]) - min(dx[:dim]) > 1e-12 * max(dx[:dim]):         # one volume size per patch update: as solvers.SubcellLimiter, refuse what would be silently wrong
        raise ValueError("limited_step: the FV patch update takes one volume size, dx = %s" % (list(dx),))
    for idx in zip(*np.nonzero(mask)):
        patch = np.zeros((S,) * dim + (u.shape[-1],))
        core = (slice(1, -1),) * dim
        patch[core] = proj[idx]
        # fill every halo entry with the nearest interior value first (edges/corners are never read)
        padded = np.pad(proj[idx], [(1, 1)] * dim + [(0, 0)], mode="edge")
        patch[...] = padded
        for a in range(dim):
            for side, off in ((0, -1), (1, +1)):
                nb = list(idx)
                nb[a] = (nb[a] + off) % u.shape[a]
                layer = np.take(proj[tuple(nb)], Ns - 1 if side == 0 else 0, axis=a)
                sl = [slice(1, -1)] * dim
                sl[a] = 0 if side == 0 else S - 1
                patch[tuple(sl)] = layer
        patch = fv_update(patch, dt, dx[0] / Ns)
        unew[idx] = apply_all_axes(R, patch[core], dim, 0)
    return unew
