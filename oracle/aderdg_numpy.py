"""ORACLE (test infrastructure only -- never imported by the product path).

Vectorised numpy restatement of one ADER-DG step (space-time predictor Picard
loop, time averages, volume integral, face extrapolation, Rusanov Riemann
solve, surface corrector) on a periodic regular Cartesian grid, d in {1,2,3}.

Reference anchor: none -- /root/reference holds no ADER-DG code (SURVEY.md
F2).  Follows SURVEY.md Appendix A.2-A.4.  "parity unpinned" against the
reference; pinned by the KATs of SURVEY.md A.5 (tests/test_aderdg_kat.py).
The point-wise Euler terms follow the arithmetic of the reference's
`Unit test/Functions.cpp:9-62` (gamma = 1.4, conservative variables), extended
to a 3-component momentum (rho, m0, m1, m2, E) and without the stray
`F[3]` overwrite of its 3-D branch.

Array layout: u[grid (d axes), node (d axes), var]; axis 0 of the grid/node
index is the reference's `i` (normal = 0), the slowest one.
"""
import numpy as np
from .dg_operators import operators

GAMMA = 1.4


class Advection:
    """Linear advection of m variables with velocity a (length d)."""

    def __init__(self, a, m=1):
        self.a = np.asarray(a, dtype=float)
        self.m = m

    def flux(self, q, d):
        return self.a[d] * q

    def maxeig(self, q, d):
        return np.full(q.shape[:-1], abs(self.a[d]))


class Euler:
    """Compressible Euler, 5 variables (rho, m0, m1, m2, E), gamma = 1.4."""
    m = 5

    def flux(self, q, d):
        rho = q[..., 0]
        irho = 1.0 / rho
        e = q[..., 4]
        p = (GAMMA - 1) * (e - 0.5 * irho * (q[..., 1] * q[..., 1] + q[..., 2] * q[..., 2] + q[..., 3] * q[..., 3]))
        coeff = irho * q[..., d + 1]
        F = np.empty_like(q)
        F[..., 0] = coeff * rho
        F[..., 1] = coeff * q[..., 1]
        F[..., 2] = coeff * q[..., 2]
        F[..., 3] = coeff * q[..., 3]
        F[..., 4] = coeff * e + coeff * p
        F[..., d + 1] += p
        return F

    def maxeig(self, q, d):
        irho = 1.0 / np.abs(q[..., 0])
        e = q[..., 4]
        p = (GAMMA - 1) * (e - 0.5 * irho * (q[..., 1] * q[..., 1] + q[..., 2] * q[..., 2] + q[..., 3] * q[..., 3]))
        c = np.sqrt(GAMMA * np.abs(p) * irho)
        un = q[..., d + 1] * irho
        return np.maximum(np.abs(un - c), np.abs(un + c))


def _apply(M, A, axis):
    """out[.., i, ..] = sum_j M[i][j] A[.., j, ..] along `axis`."""
    return np.moveaxis(np.tensordot(M, A, axes=([1], [axis])), 0, axis)


def _dim(u):
    return (u.ndim - 1) // 2


def predictor(u, dt, dx, ops, pde, n_it=None):
    """Space-time predictor q[grid, l, node, var] (A.2)."""
    d = _dim(u)
    N = ops['N']
    if n_it is None:
        n_it = N
    D, iK1, F0, w = ops['D'], ops['iK1'], ops['F0'], ops['w']
    ue = np.expand_dims(u, d)                      # time axis at position d
    q = np.repeat(ue, N, axis=d)
    tshape = [1] * q.ndim
    tshape[d] = N
    for _ in range(n_it):
        S = np.zeros_like(q)
        for a in range(d):
            S += _apply(D, pde.flux(q, a), d + 1 + a) / dx[a]
        if hasattr(pde, "source"):                 # q_t + div F = S(q): the algebraic source enters beside the flux divergence
            S -= pde.source(q)
        R = F0.reshape(tshape) * ue - dt * w.reshape(tshape) * S
        q = _apply(iK1, R, d)
    return q


def time_averages(q, ops, pde):
    d = (q.ndim - 2) // 2
    w = ops['w']
    qbar = np.tensordot(w, q, axes=([0], [d]))
    Fbar = [np.tensordot(w, pde.flux(q, a), axes=([0], [d])) for a in range(d)]
    return qbar, Fbar


def volume(u, Fbar, dt, dx, ops):
    d = _dim(u)
    N = ops['N']
    Kxi, w = ops['Kxi'], ops['w']
    us = u.copy()
    for a in range(d):
        sh = [1] * u.ndim
        sh[d + a] = N
        us += dt / dx[a] * _apply(Kxi, Fbar[a], d + a) / w.reshape(sh)
    return us


def traces(qbar, Fbar, ops):
    """Return lists (per direction a) of qL,qR,FL,FR with the node axis a removed."""
    d = _dim(qbar)
    out = []
    for a in range(d):
        ax = d + a
        qL = np.tensordot(ops['phiL'], qbar, axes=([0], [ax]))
        qR = np.tensordot(ops['phiR'], qbar, axes=([0], [ax]))
        FL = np.tensordot(ops['phiL'], Fbar[a], axes=([0], [ax]))
        FR = np.tensordot(ops['phiR'], Fbar[a], axes=([0], [ax]))
        # tensordot puts the contracted result's remaining axes in order -> axis removed
        out.append((qL, qR, FL, FR))
    return out


def riemann(tr, pde, d):
    """Rusanov flux at the RIGHT face of every cell, per direction (periodic).

    Fstar[a][grid..., transverse nodes..., var] is the flux between cell c (its
    R trace) and cell c+e_a (its L trace); s = max over the face's nodes (A.4).
    """
    Fs = []
    for a in range(d):
        qL, qR, FL, FR = tr[a]
        qm, Fm = qR, FR                                   # "-" side: this cell's R
        qp, Fp = np.roll(qL, -1, axis=a), np.roll(FL, -1, axis=a)   # "+": right neighbour's L
        lam = np.maximum(pde.maxeig(qm, a), pde.maxeig(qp, a))
        if d > 1:
            s = lam.max(axis=tuple(range(d, d + d - 1)), keepdims=True)
        else:
            s = lam
        Fs.append(0.5 * (Fm + Fp) - 0.5 * s[..., None] * (qp - qm))
    return Fs


def corrector(us, Fstar, dt, dx, ops):
    d = _dim(us)
    N = ops['N']
    w, phiL, phiR = ops['w'], ops['phiL'], ops['phiR']
    un = us.copy()
    for a in range(d):
        FR = np.expand_dims(Fstar[a], d + a)                      # right face of cell
        FLft = np.expand_dims(np.roll(Fstar[a], 1, axis=a), d + a)  # left face = left neighbour's right face
        sh = [1] * us.ndim
        sh[d + a] = N
        un -= dt / dx[a] * (phiR.reshape(sh) * FR - phiL.reshape(sh) * FLft) / w.reshape(sh)
    return un


def step(u, dt, dx, ops, pde, n_it=None, stages=False):
    d = _dim(u)
    q = predictor(u, dt, dx, ops, pde, n_it)
    qbar, Fbar = time_averages(q, ops, pde)
    us = volume(u, Fbar, dt, dx, ops)
    if hasattr(pde, "source"):                     # + dt * time average of the source over the predictor
        us = us + dt * np.tensordot(ops['w'], pde.source(q), axes=([0], [d]))
    tr = traces(qbar, Fbar, ops)
    Fs = riemann(tr, pde, d)
    un = corrector(us, Fs, dt, dx, ops)
    if stages:
        return dict(q=q, qbar=qbar, Fbar=Fbar, ustar=us, traces=tr, Fstar=Fs, unew=un)
    return un


def step_single_stage(u, dt, dx, ops, pde):
    """cfg 1 variant: volume + Riemann + corrector only (qbar := u, Fbar := f(u))."""
    d = _dim(u)
    Fbar = [pde.flux(u, a) for a in range(d)]
    us = volume(u, Fbar, dt, dx, ops)
    if hasattr(pde, "source"):
        us = us + dt * pde.source(u)
    tr = traces(u, Fbar, ops)
    Fs = riemann(tr, pde, d)
    return corrector(us, Fs, dt, dx, ops)


def node_coords(grid, N, ops):
    """Physical coordinates on [0,1]^d: list of arrays broadcastable to grid+nodes."""
    d = len(grid)
    xs = []
    for a in range(d):
        h = 1.0 / grid[a]
        x = (np.arange(grid[a])[:, None] + ops['xi'][None, :]) * h      # [cells, N]
        sh = [1] * (2 * d)
        sh[a] = grid[a]
        sh[d + a] = N
        xs.append(x.reshape(sh))
    return xs


# ---------------------------------------------------------------------------------------------------------------------
# Terms that depend on position / time, non-conservative product:  q_t + div F(q, x, t) + B(q, x, t) . grad q = S(q, x, t)
# (the hooks `Unit test/correctness_test.cpp:16-41,145-155` declares; no ADER-DG in the reference: parity unpinned).
# pde: flux(q, x, t, a), maxeig(q, x, t, a), optional source(q, x, t), ncp(q, dq, x, t, a); x = list of d arrays broadcastable
# to q[..., 0].  Restated by exahype_amd/csrc/exa_dg_plain.hpp (stage A) and dg_stage_b_dense_kernel (stage B).
# ---------------------------------------------------------------------------------------------------------------------
def _coords(grid, N, ops, dx, origin, lead=0):
    """node coordinates: list over axes of arrays shaped [grid..., (lead singleton axes), nodes...] with 1 where they do not vary"""
    d = len(grid)
    xs = []
    for a in range(d):
        x = origin[a] + (np.arange(grid[a])[:, None] + ops['xi'][None, :]) * dx[a]      # [cells, N]
        sh = [1] * (2 * d + lead)
        sh[a] = grid[a]
        sh[d + lead + a] = N
        xs.append(x.reshape(sh))
    while len(xs) < 3:
        xs.append(np.zeros([1] * (2 * d + lead)))
    return xs


def step_xt(u, dt, dx, ops, pde, t=0.0, origin=None, n_it=None, stages=False):
    d = _dim(u)
    N = ops['N']
    grid = u.shape[:d]
    origin = [0.0] * d if origin is None else origin
    if n_it is None:
        n_it = N
    D, iK1, F0, w, xi = ops['D'], ops['iK1'], ops['F0'], ops['w'], ops['xi']
    has_src, has_ncp = hasattr(pde, "source"), hasattr(pde, "ncp")
    x4 = _coords(grid, N, ops, dx, origin, lead=1)                  # with a time axis at position d
    tshape = [1] * (2 * d + 1)
    tshape[d] = N
    tl = (t + xi * dt).reshape(tshape)
    ue = np.expand_dims(u, d)
    q = np.repeat(ue, N, axis=d)
    tsv = tshape + [1]

    def ncp_sum(qq):
        out = np.zeros_like(qq)
        for a in range(d):
            out += pde.ncp(qq, _apply(D, qq, d + 1 + a) / dx[a], x4, tl, a)
        return out

    for _ in range(n_it):
        S = np.zeros_like(q)
        for a in range(d):
            S += _apply(D, pde.flux(q, x4, tl, a), d + 1 + a) / dx[a]
        if has_ncp:
            S += ncp_sum(q)
        if has_src:
            S -= pde.source(q, x4, tl)
        R = F0.reshape(tsv) * ue - dt * w.reshape(tsv) * S
        q = _apply(iK1, R, d)
    qbar = np.tensordot(w, q, axes=([0], [d]))
    Fbar = [np.tensordot(w, pde.flux(q, x4, tl, a), axes=([0], [d])) for a in range(d)]
    us = volume(u, Fbar, dt, dx, ops)
    if has_ncp:
        us = us - dt * np.tensordot(w, ncp_sum(q), axes=([0], [d]))
    if has_src:
        us = us + dt * np.tensordot(w, pde.source(q, x4, tl), axes=([0], [d]))
    tr = traces(qbar, Fbar, ops)
    # Riemann solve at the RIGHT face of every cell, at the face nodes and t + dt / 2; with an ncp the jump term D goes half to either side
    x3 = _coords(grid, N, ops, dx, origin)
    tf = t + 0.5 * dt
    un = us.copy()
    for a in range(d):
        qL, qR, FL, FR = tr[a]
        xf = []
        for b in range(3):
            if b >= d:
                xf.append(np.zeros([1] * (2 * d - 1)))
            elif b == a:
                sh = [1] * (2 * d - 1)
                sh[a] = grid[a]
                xf.append((origin[a] + (np.arange(grid[a]) + 1.0) * dx[a]).reshape(sh))
            else:
                xf.append(np.squeeze(x3[b], axis=d + a))               # node axis a removed
        qm, Fm = qR, FR
        qp, Fp = np.roll(qL, -1, axis=a), np.roll(FL, -1, axis=a)
        lam = np.maximum(pde.maxeig(qm, xf, tf, a) * np.ones(qm.shape[:-1]), pde.maxeig(qp, xf, tf, a) * np.ones(qm.shape[:-1]))
        s = lam.max(axis=tuple(range(d, d + d - 1)), keepdims=True) if d > 1 else lam
        Fs = 0.5 * (Fm + Fp) - 0.5 * s[..., None] * (qp - qm)
        Dj = pde.ncp(0.5 * (qm + qp), qp - qm, xf, tf, a) if has_ncp else np.zeros_like(Fs)
        # the cell left of the face (its high face) takes F* + D/2, the cell right of it (its low face) F* - D/2.  The right cell evaluates
        # the face at ITS low-face coordinates: the same point, except across the periodic wrap
        xlo = list(xf)
        sh = [1] * (2 * d - 1)
        sh[a] = grid[a]
        xlo[a] = (origin[a] + np.arange(grid[a]) * dx[a]).reshape(sh)
        qm_l, qp_l = np.roll(qR, 1, axis=a), qL                       # low face of every cell: minus = left neighbour's R, plus = own L
        lam_l = np.maximum(pde.maxeig(qm_l, xlo, tf, a) * np.ones(qL.shape[:-1]), pde.maxeig(qp_l, xlo, tf, a) * np.ones(qL.shape[:-1]))
        s_l = lam_l.max(axis=tuple(range(d, d + d - 1)), keepdims=True) if d > 1 else lam_l
        Flo = 0.5 * (np.roll(FR, 1, axis=a) + FL) - 0.5 * s_l[..., None] * (qp_l - qm_l)
        if has_ncp:
            Flo = Flo - 0.5 * pde.ncp(0.5 * (qm_l + qp_l), qp_l - qm_l, xlo, tf, a)
        Fhi = Fs + 0.5 * Dj
        sh = [1] * us.ndim
        sh[d + a] = N
        un -= dt / dx[a] * (ops['phiR'].reshape(sh) * np.expand_dims(Fhi, d + a) - ops['phiL'].reshape(sh) * np.expand_dims(Flo, d + a)) / w.reshape(sh)
    if stages:
        return dict(q=q, qbar=qbar, Fbar=Fbar, ustar=us, traces=tr, unew=un)
    return un

