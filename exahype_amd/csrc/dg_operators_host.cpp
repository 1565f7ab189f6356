// Reference-element operators for the ADER-DG kernels,
// computed on the host in long double when a plan is created.  No counterpart
// in the reference (it has no quadrature/basis code).
#include <cmath>
#include "exa_launch.hpp"

namespace exa {

typedef long double ld;

// Legendre P_n(x) and derivative by the three-term recurrence
static void legendre(int n, ld x, ld* p, ld* dp) {
    ld p0 = 1.0L, p1 = x;
    if (n == 0) { *p = 1.0L; *dp = 0.0L; return; }
    for (int k = 2; k <= n; k++) {
        ld pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
        p0 = p1;
        p1 = pk;
    }
    *p = p1;
    *dp = n * (x * p1 - p0) / (x * x - 1.0L);
}

int build_dg_operators(int N, DgOpsHost* o) {
    if (N < 1 || N > MAXN) return -1;
    o->N = N;
    ld x[MAXN], w[MAXN];
    const ld pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < N; i++) {
        // Chebyshev guess, Newton on P_N; ascending order
        ld z = -cosl(pi * (i + 0.75L) / (N + 0.5L));
        for (int it = 0; it < 100; it++) {
            ld p, dp;
            legendre(N, z, &p, &dp);
            ld dz = p / dp;
            z -= dz;
            if (fabsl(dz) < 1e-19L) break;
        }
        ld p, dp;
        legendre(N, z, &p, &dp);
        x[i] = 0.5L * (z + 1.0L);                       // map [-1,1] -> [0,1]
        w[i] = 1.0L / ((1.0L - z * z) * dp * dp);        // = (2/((1-z^2) dp^2)) / 2
    }
    // barycentric weights and derivative matrix D[i][j] = phi_j'(x_i)
    ld bw[MAXN];
    for (int j = 0; j < N; j++) {
        bw[j] = 1.0L;
        for (int k = 0; k < N; k++)
            if (k != j) bw[j] /= (x[j] - x[k]);
    }
    ld D[MAXN][MAXN];
    for (int i = 0; i < N; i++) {
        ld s = 0.0L;
        for (int j = 0; j < N; j++) {
            if (i == j) continue;
            D[i][j] = (bw[j] / bw[i]) / (x[i] - x[j]);
            s += D[i][j];
        }
        D[i][i] = -s;
    }
    // boundary values of the Lagrange basis
    ld pL[MAXN], pR[MAXN];
    for (int j = 0; j < N; j++) {
        pL[j] = 1.0L;
        pR[j] = 1.0L;
        for (int k = 0; k < N; k++)
            if (k != j) {
                pL[j] *= (0.0L - x[k]) / (x[j] - x[k]);
                pR[j] *= (1.0L - x[k]) / (x[j] - x[k]);
            }
    }
    // Kxi[i][j] = w_j D[j][i];  K1 = phiR phiR^T - Kxi;  iK1 by Gauss-Jordan with partial pivoting
    ld K1[MAXN][MAXN], A[MAXN][2 * MAXN];
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            ld kxi = w[j] * D[j][i];
            o->Kxi[i * N + j] = (double)kxi;
            K1[i][j] = pR[i] * pR[j] - kxi;
            o->K1[i * N + j] = (double)K1[i][j];
            A[i][j] = K1[i][j];
            A[i][N + j] = (i == j) ? 1.0L : 0.0L;
        }
    for (int c = 0; c < N; c++) {
        int piv = c;
        for (int r = c + 1; r < N; r++)
            if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0L) return -1;
        if (piv != c)
            for (int k = 0; k < 2 * N; k++) { ld t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        ld inv = 1.0L / A[c][c];
        for (int k = 0; k < 2 * N; k++) A[c][k] *= inv;
        for (int r = 0; r < N; r++) {
            if (r == c) continue;
            ld f = A[r][c];
            if (f == 0.0L) continue;
            for (int k = 0; k < 2 * N; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int i = 0; i < N; i++) {
        o->xi[i] = (double)x[i];
        o->w[i] = (double)w[i];
        o->phiL[i] = (double)pL[i];
        o->phiR[i] = (double)pR[i];
        for (int j = 0; j < N; j++) {
            o->D[i * N + j] = (double)D[i][j];
            o->iK1[i * N + j] = (double)A[i][N + j];
        }
    }
    return 0;
}

// FV subcell limiter: P[s][i] = Ns * int_{s/Ns}^{(s+1)/Ns} phi_i (Gauss-Legendre on every
// subinterval: exact), R = argmin ||P u - v||^2 s.t. w.u = mean(v) (KKT system, Gauss-Jordan in long double).
int build_limiter_operators(const DgOpsHost* o, int Ns, double* P, double* R) {
    const int N = o->N;
    if (Ns < 1 || Ns > 2 * MAXN) return -1;
    ld Pm[2 * MAXN][MAXN];
    for (int s = 0; s < Ns; s++)
        for (int i = 0; i < N; i++) Pm[s][i] = 0.0L;
    for (int s = 0; s < Ns; s++)
        for (int k = 0; k < N; k++) {
            const ld x = ((ld)s + (ld)o->xi[k]) / Ns;              // the GL nodes/weights of the element serve as the rule
            for (int i = 0; i < N; i++) {
                ld phi = 1.0L;
                for (int m = 0; m < N; m++)
                    if (m != i) phi *= (x - (ld)o->xi[m]) / ((ld)o->xi[i] - (ld)o->xi[m]);
                Pm[s][i] += (ld)o->w[k] * phi;
            }
        }
    for (int s = 0; s < Ns; s++)
        for (int i = 0; i < N; i++) P[s * N + i] = (double)Pm[s][i];
    // KKT: [2 P^T P, w; w^T, 0] [u; lambda] = [2 P^T v; mean(v)]
    const int M = N + 1;
    ld K[MAXN + 1][MAXN + 1 + 2 * MAXN];
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M + Ns; j++) K[i][j] = 0.0L;
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < N; j++) {
            ld a = 0.0L;
            for (int s = 0; s < Ns; s++) a += Pm[s][i] * Pm[s][j];
            K[i][j] = 2.0L * a;
        }
        K[i][N] = (ld)o->w[i];
        K[N][i] = (ld)o->w[i];
        for (int s = 0; s < Ns; s++) K[i][M + s] = 2.0L * Pm[s][i];
    }
    for (int s = 0; s < Ns; s++) K[N][M + s] = 1.0L / Ns;
    for (int c = 0; c < M; c++) {
        int piv = c;
        for (int r = c + 1; r < M; r++)
            if (fabsl(K[r][c]) > fabsl(K[piv][c])) piv = r;
        if (K[piv][c] == 0.0L) return -1;
        if (piv != c)
            for (int k = 0; k < M + Ns; k++) { ld t = K[c][k]; K[c][k] = K[piv][k]; K[piv][k] = t; }
        const ld inv = 1.0L / K[c][c];
        for (int k = 0; k < M + Ns; k++) K[c][k] *= inv;
        for (int r = 0; r < M; r++) {
            if (r == c) continue;
            const ld f = K[r][c];
            if (f == 0.0L) continue;
            for (int k = 0; k < M + Ns; k++) K[r][k] -= f * K[c][k];
        }
    }
    for (int i = 0; i < N; i++)
        for (int s = 0; s < Ns; s++) R[i * Ns + s] = (double)K[i][M + s];
    return 0;
}

}  // namespace exa
