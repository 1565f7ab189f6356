"""ORACLE (test infrastructure only -- never imported by the product path).

One-dimensional ADER-DG reference-element operators on Gauss-Legendre nodes.

Reference anchor: none.  /root/reference contains no ADER-DG code at all
(SURVEY.md F2); these follow SURVEY.md Appendix A.1 (standard nodal ADER-DG,
Dumbser et al. 2008).  "parity unpinned" against the reference; pinned instead
by the operator identities of SURVEY.md A.5 (tests/test_dg_operators.py).
"""
import numpy as np


def gauss_legendre_01(N):
    """Nodes/weights on [0,1] (sum w = 1)."""
    x, w = np.polynomial.legendre.leggauss(N)
    return 0.5 * (x + 1.0), 0.5 * w


def lagrange_eval(nodes, x):
    """phi_j(x) for all j (1-D array)."""
    N = len(nodes)
    out = np.ones(N)
    for j in range(N):
        for k in range(N):
            if k != j:
                out[j] *= (x - nodes[k]) / (nodes[j] - nodes[k])
    return out


def derivative_matrix(nodes):
    """D[i][j] = phi_j'(xi_i) via barycentric weights."""
    N = len(nodes)
    bw = np.ones(N)
    for j in range(N):
        for k in range(N):
            if k != j:
                bw[j] /= (nodes[j] - nodes[k])
    D = np.zeros((N, N))
    for i in range(N):
        for j in range(N):
            if i != j:
                D[i, j] = (bw[j] / bw[i]) / (nodes[i] - nodes[j])
        D[i, i] = -np.sum(D[i, :])
    return D


def operators(N):
    """Return dict of the A.1 operators for polynomial order p = N-1."""
    xi, w = gauss_legendre_01(N)
    D = derivative_matrix(xi)
    Kxi = D.T * w[None, :]          # Kxi[i][j] = w_j * D[j][i]
    phiL = lagrange_eval(xi, 0.0)
    phiR = lagrange_eval(xi, 1.0)
    K1 = np.outer(phiR, phiR) - Kxi
    iK1 = np.linalg.inv(K1)
    return dict(N=N, xi=xi, w=w, D=D, Kxi=Kxi, phiL=phiL, phiR=phiR,
                F0=phiL.copy(), K1=K1, iK1=iK1)
