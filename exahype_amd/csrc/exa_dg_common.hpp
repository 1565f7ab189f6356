// Shared compile-time geometry and the reference-element operator block for the
// ADER-DG kernels (no counterpart in the reference).
#pragma once
#include <hip/hip_runtime.h>

namespace exa {

constexpr int MAXN = 8;

__host__ __device__ constexpr int ipow(int b, int e) { return e <= 0 ? 1 : b * ipow(b, e - 1); }

// 1-D operators, passed BY VALUE as a kernel argument: the kernarg segment is
// read with scalar loads, so every matrix entry becomes an SGPR operand of
// v_fma_f64 and costs no VGPR / LDS traffic.
template <int N> struct DgOps {
    double w[N];        // Gauss-Legendre weights on [0,1]
    double iw[N];       // 1 / w
    double D[N * N];    // D[i][j]   = phi_j'(xi_i)
    double DT[N * N];   // DT[j][i]  = D[i][j]  (column j of D contiguous)
    // even-odd form of D (centro-antisymmetric: D[N-1-i][N-1-j] = -D[i][j]), h = N/2, j < h, jb = N-1-j:
    //   DEO[j][i]     = (D[i][j] + D[i][jb]) / 2   acts on F_j + F_jb:  s_i - s_{N-1-i} = 2 sum_j (..)   (i < h)
    //   DEO[j][h + i] = (D[i][j] - D[i][jb]) / 2   acts on F_j - F_jb:  s_i + s_{N-1-i} = 2 sum_j (..)
    // odd N (middle index h): DEO[j][2h] = D[h][j] (the middle row sees only F_j - F_jb; D[h][h] = 0) and
    //   DEO[h][i] = D[i][h] (the middle node's flux enters s_i - s_{N-1-i} with 2 D[i][h])
    double DEO[(N / 2) * N + N / 2 + 1];
    double Kxi[N * N];  // Kxi[i][j] = w_j D[j][i]
    double KEO[(N / 2) * N + N / 2 + 1];   // even-odd form of Kxi (centro-antisymmetric like D: w is symmetric), same packing as DEO
    double T[N * N];    // T[l'][l]  = iK1[l'][l] * w_l   (time update; uses iK1*F0 = 1)
    double Tsum[N];     // sum_l T[l'][l]                  (iteration 0: the iterate is constant in time)
    double phiL[N];
    double phiR[N];
};

// LDS image of one cell: SoA [var][time slab][node], node index lexicographic (i slowest).
// 3-D: unpadded -- scripts/lds_stride_search.py shows a padded i-stride makes the k-pencils
// (stride-N*N+pad lanes) up to 6-way conflicted, while N*N keeps them conflict-free as
// ds_read_b128 and leaves only a 2-way conflict on the j-pencils (MI355X_MICROARCH.md LDS table).
template <int DIM, int N> struct Geo {
    static constexpr int NN = ipow(N, DIM);       // nodes per cell
    static constexpr int NF = ipow(N, DIM - 1);   // nodes per face
    static constexpr int PAD = (DIM == 3) ? 0 : 1;
    static constexpr int SI = ((DIM == 3) ? N * N : N) + PAD;  // stride of axis 0
    static constexpr int SL = N * SI;                           // one (var, slab) image
    static constexpr int NTS = (N > DIM + 1) ? N : DIM + 1;     // slabs per variable

    // padded offset of lexicographic node n
    __host__ __device__ static inline int node_off(int n) {
        if constexpr (DIM == 3) return (n / (N * N)) * SI + (n % (N * N));
        else return (n / N) * SI + (n % N);
    }
    // axis-0 coordinate etc. of node n
    __device__ static inline int coord(int n, int d) {
        if constexpr (DIM == 3) return d == 0 ? n / (N * N) : (d == 1 ? (n / N) % N : n % N);
        else return d == 0 ? n / N : n % N;
    }
    // stride between consecutive nodes of a pencil along axis d
    __host__ __device__ static constexpr int pstride(int d) {
        if constexpr (DIM == 3) return d == 0 ? SI : (d == 1 ? N : 1);
        else return d == 0 ? SI : 1;
    }
    // padded offset of the first node of pencil t (t = lexicographic index of the
    // remaining axes == the face-node index y of the traces)
    __host__ __device__ static inline int pbase(int d, int t) {
        if constexpr (DIM == 3) {
            if (d == 0) return t;                          // (j,k)
            if (d == 1) return (t / N) * SI + (t % N);     // (i,k)
            return (t / N) * SI + (t % N) * N;             // (i,j)
        } else {
            return d == 0 ? t : t * SI;
        }
    }
    // face-node index y (for direction d) of lexicographic node n
    __device__ static inline int face_index(int n, int d) {
        if constexpr (DIM == 3) {
            const int i = n / (N * N), j = (n / N) % N, k = n % N;
            return d == 0 ? j * N + k : (d == 1 ? i * N + k : i * N + j);
        } else {
            return d == 0 ? n % N : n / N;
        }
    }
};

// Where and when, for term sets whose terms depend on position / time (`Unit test/correctness_test.cpp:16-41`: flux / maxEigenvalue / sourceTerm
// (Q, x, h, t, dt, ...)): node x = x0 + (cell + xi_i) h per axis, level time t_l = t + xi_l dt.  A kernel argument (scalar registers); the built-in
// term sets never look at it.
struct PlainGeo {
    double x0[3];        // physical coordinates of the block's origin
    double h[3];         // cell size
    double t;            // time at the start of the step
    double xi[MAXN];     // Gauss-Legendre nodes on [0, 1]
};
// xi[i] for a run-time i without indexing the argument dynamically (a select chain over scalar registers)
template <int N> __device__ inline double xi_of(const PlainGeo& g, int i) {
    double r = g.xi[0];
#pragma unroll
    for (int k = 1; k < N; k++) r = i == k ? g.xi[k] : r;
    return r;
}

// what the one-kernel step (exa_dg_reg.hpp, FUSE) needs of the previous step
struct RegFuse {
    const double* trace_in;       // traces of the previous step
    const double* ghost[6];       // neighbour blocks' traces at the block faces (or null: periodic)
    double dt_prev;               // the previous step's dt (its corrector)
    double* u_plain;              // where the corrected u of the previous step goes as well, or null
};

// A box [lo, lo+nb) of cells inside the local block nc[3] (nc[2] = nb[2] = 1 in 2-D);
// box slots are enumerated lexicographically, last axis fastest.
struct CellBox {
    long nc[3], lo[3], nb[3], nbox;
    __host__ __device__ inline long cell(long b) const {
        if (b >= nbox) return -1;
        const long cz = b % nb[2];
        b /= nb[2];
        const long cy = b % nb[1], cx = b / nb[1];
        return ((lo[0] + cx) * nc[1] + lo[1] + cy) * nc[2] + lo[2] + cz;
    }
};

}  // namespace exa
