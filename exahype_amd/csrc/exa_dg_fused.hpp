// Fused single-stage ADER-DG step in 2-D (n_picard = 0: qbar := u, Fbar := f(u) -- BASELINE configs[1], "volume +
// Riemann only"): volume integral, face extrapolation, Rusanov flux and surface corrector in ONE launch, so the face
// traces never travel through HBM.  Same scheme and results (to rounding) as dg_stage_a_single_kernel followed by
// dg_stage_b_kernel; no counterpart in the reference.
//
// One workgroup per tile of TX x TY cells plus the ring of face neighbours (periodic wrap), all in LDS:
//   A  load      u of the tile and its ring, AoS -> [cell][var][node]                         (coalesced)
//   B  pencils   per (cell, direction, transverse node): fluxes at the N nodes, the four traces q/F at L/R; the L trace (the
//                plus side of the cell's L face) -> face table, the R trace stays in registers; trace eigenvalues -> table;
//                interior cells also the volume term (kept in registers, added direction by direction)
//   C  faces     the pencil lane on the MINUS side of a face: face-wide max of the eigenvalues (no shuffle, any N), Rusanov
//                flux F* from its own R trace and the plus side's table entry -> table; then the pencils of the interior
//                cells apply the lift (F* of the R face from registers, of the L face from the table)
//   D  store     u of the tile                                                                 (coalesced)
// HBM traffic per cell: (1 + ring/tile) reads + 1 write of its DoF (1 920 B at p = 3 with 4 x 4 tiles before L2 hits on
// the shared rings; 6 400 B for the two-kernel path, 3 840 B for the survey's fused ideal with traces in HBM).
// Needs separate input and output arrays (neighbouring tiles read what this one would overwrite).
#pragma once
#include "exa_dg_kernels.hpp"

// 8-byte LDS load that the back-end cannot pair into ds_read2_b64 (half the LDS rate of two ds_read_b64)
#define EXA_FLD(ptr) (*(const volatile __attribute__((address_space(3))) double*)(ptr))

namespace exa {

template <int N> struct FusedTile { static constexpr int T = N <= 4 ? 4 : (N <= 6 ? 3 : 2); };

template <int N, class PDE, int TX, int TY> struct FusedSingle {
    static constexpr int NV = PDE::NV, NN = N * N;
    static constexpr int LX = TX + 2, LY = TY + 2, LC = LX * LY;
    static constexpr int CS = NV * NN;                          // doubles per cell in HBM
    // LDS image of a cell (r3): the HBM order [node][var] at a cell stride CSP = 4 mod 8, the four unused corner cells of the ring left out
    // (slot()).  The first layout ([var][node], cell stride NV N^2 = 80 = 16 mod 32) was transposed on the way in and out (5-way conflicted
    // stores at a 16-double stride) and its pencil reads were 4-way conflicted: 72 % of the LDS cycles were bank conflicts.  Now the load
    // and the store are plain copies, and a 32-lane group of x-pencil tasks -- N pencils x 32 / N cells -- reads distinct banks: the lanes
    // of a cell are NV apart, cells whose slots differ mod 8 sit on the 8 multiples of 4; the lane order pairs the tile's cell rows (0, 2)
    // and (1, 3) for that.  (The y pencils of a cell are N NV = 20 apart, multiples of 4 like the cells: still 4-way.  A row pitch of N + 1
    // nodes makes them conflict-free too, but the image then no longer leaves room for three workgroups per CU: 0.243 against 0.204 ms; the four
    // padding doubles of a cell spread one behind each row -- row stride 21, same size -- frees them as well but costs more in the copies,
    // whose odd rows lose their 16-byte alignment: 0.177 against 0.170 ms.)
    static constexpr int RS = N * NV;                           // row stride
    static constexpr int CSP = N * RS + ((4 - (N * RS) % 8) + 8) % 8;
    static constexpr int LCS = LX * LY - 4;                     // cell slots
    __host__ __device__ static constexpr int slot(int lc) { return lc - (lc < LY - 1 ? 1 : (lc < LX * LY - LY ? 2 : 3)); }
    static constexpr int NFX = (TX + 1) * TY, NFY = TX * (TY + 1), NFACE = NFX + NFY;
    // per face: [field q|F][var][node] of its PLUS side (the minus side's pencil lane keeps its own trace in registers and computes the Rusanov
    // flux, r4: half the table -- 37 KB per workgroup at p = 3 -- and no separate face tasks)
    static constexpr int FS0 = 2 * NV * N;
    static constexpr int FS = FS0 + ((4 - FS0 % 8) + 8) % 8;    // (stride = 4 mod 8, as the cells)
    static constexpr int LAMO = LCS * CSP + NFACE * FS;         // offset of the eigenvalue table [face][side][node]
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)(LAMO + NFACE * 2 * N);
    static constexpr int T_INT = TX * TY * 2 * N, T_HX = 2 * TY * N, T_HY = 2 * TX * N;
    static constexpr int NT = 256;
    static_assert(T_INT + T_HX + T_HY <= NT, "one pencil task per lane");
};

template <int N, class PDE, int TX, int TY>
__global__ void __launch_bounds__(256)
dg_fused_single_kernel(const double* __restrict__ u_in, double* __restrict__ u_out, long ncx, long ncy, long tiles_y, double dt,
                       double idx0, double idx1, const void* __restrict__ ops_raw) {
    using FU = FusedSingle<N, PDE, TX, TY>;
    constexpr int NV = PDE::NV, LY = FU::LY, LC = FU::LC, CS = FU::CS, CSP = FU::CSP, RS = FU::RS;
    constexpr int NFX = FU::NFX, NFACE = FU::NFACE, FS = FU::FS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* U = lds;
    double* FT = lds + FU::LCS * CSP;

    double* LAM = lds + FU::LAMO;
    const int tid = threadIdx.x;
    // neighbouring tiles (shared rings) on the same XCD's L2 (xcd_contiguous); 32-bit arithmetic: the grid is the tile count, and a 64-bit
    // division is ~100 scalar instructions in front of the tile's first load
    const unsigned nb = gridDim.x, bi = blockIdx.x, per = nb / 8u;
    const unsigned lb = bi < per * 8u ? (bi % 8u) * per + bi / 8u : bi;
    const unsigned ty_n = (unsigned)tiles_y, lbx = lb / ty_n;
    const long tx0 = (long)lbx * TX, ty0 = (long)(lb - lbx * ty_n) * TY;
    const double idx[2] = {idx0, idx1};
    const bool wide = CS % 2 == 0 && CSP % 2 == 0 && ((reinterpret_cast<unsigned long long>(u_in) | reinterpret_cast<unsigned long long>(u_out)) & 15) == 0;

    // ---- A: load the tile and its ring (corners are never used); global cell of every local cell first (one lane
    //      each: the periodic wrap costs 64-bit remainders, far too slow to repeat per element)
    __shared__ long gcell[LC];
    if (tid < LC) {
        const int lx = tid / LY - 1, ly = tid % LY - 1;
        long g = -1;
        if (!((lx < 0 || lx >= TX) && (ly < 0 || ly >= TY))) {
            long gx = tx0 + lx, gy = ty0 + ly;                   // periodic wrap by comparison (a 64-bit remainder is ~100 instructions);
            gx = gx < 0 ? gx + ncx : gx;                         // a ring cell of a partial tile at the end of the block may lie a period further
            gy = gy < 0 ? gy + ncy : gy;
            while (gx >= ncx) gx -= ncx;
            while (gy >= ncy) gy -= ncy;
            g = gx * ncy + gy;
        }
        gcell[tid] = g;
    }
    __syncthreads();
    {
        // 16-byte copies where the cell size is even and the arrays are 16-byte aligned, 8-byte copies otherwise
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        if (wide) {
            constexpr int CP = CS / 2;
            constexpr int PER = (LC * CP + FU::NT - 1) / FU::NT; // all loads of a lane in flight before the first use
            v2d_t tmp[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int e = tid + k * FU::NT;
                const int lc = e < LC * CP ? e / CP : 0;
                const long g = gcell[lc];
                tmp[k] = (e < LC * CP && g >= 0) ? reinterpret_cast<const v2d_t*>(u_in + g * CS)[e - lc * CP] : v2d_t{0.0, 0.0};
            }
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int e = tid + k * FU::NT;
                if (e < LC * CP) {
                    const int lc = e / CP;
                    if (gcell[lc] >= 0) reinterpret_cast<v2d_t*>(U + FU::slot(lc) * CSP)[e - lc * CP] = tmp[k];
                }
            }
        } else {
            for (int e = tid; e < LC * CS; e += FU::NT) {
                const int lc = e / CS;
                const long g = gcell[lc];
                if (g >= 0) U[FU::slot(lc) * CSP + (e - lc * CS)] = u_in[g * CS + (e - lc * CS)];
            }
        }
    }
    __syncthreads();

    // ---- pencil task of this lane: interior cells both directions, ring cells the direction that faces the tile
    int p_lx = 0, p_ly = 0, p_d = 0, p_t = 0;
    bool p_task = true, p_int = false;
    {
        int k = tid;
        if (k < FU::T_INT) {                                     // direction slowest: waves are (mostly) direction-uniform
            p_int = true;
            p_t = k % N; k /= N;
            p_ly = k % TY; k /= TY;
            p_lx = k % TX; p_d = k / TX;
            if constexpr (TX == 4) p_lx = ((p_lx & 1) << 1) | (p_lx >> 1);      // rows in the order 0, 2, 1, 3 (see FusedSingle)
        } else if ((k -= FU::T_INT) < FU::T_HX) {
            p_d = 0;
            p_t = k % N; k /= N;
            p_ly = k % TY; p_lx = (k / TY) ? TX : -1;
        } else if ((k -= FU::T_HX) < FU::T_HY) {
            p_d = 1;
            p_t = k % N; k /= N;
            p_lx = k % TX; p_ly = (k / TX) ? TY : -1;
        } else {
            p_task = false;
        }
    }
    const int p_lc = FU::slot((p_lx + 1) * LY + p_ly + 1);
    const int p_n0 = p_d == 0 ? p_t * NV : p_t * RS;              // first node of the pencil (offset in the cell's image), stride p_ns
    const int p_ns = p_d == 0 ? RS : NV;
    // faces of the pencil's cell along p_d: index or -1 (outside the tile's face table)
    int fL = -1, fR = -1;
    if (p_task) {
        if (p_d == 0) {
            if (p_lx >= 0) fL = p_lx * TY + p_ly;
            if (p_lx + 1 <= TX) fR = (p_lx + 1) * TY + p_ly;
        } else {
            if (p_ly >= 0) fL = NFX + p_lx * (TY + 1) + p_ly;
            if (p_ly + 1 <= TY) fR = NFX + p_lx * (TY + 1) + p_ly + 1;
        }
    }

    // ---- B: fluxes, traces, volume term
    double vol[N][NV], qR[NV], FR[NV];                            // (R trace, then F* of the R face)
    if (p_task) {
        const EXA_AS4 DgOps<N>* o = ops_here<N>(ops_raw);
        double F[N][NV], q[N][NV], qLv[NV];
#pragma unroll
        for (int j = 0; j < N; j++) {
            double a[nz(PDE::NAUX)];
#pragma unroll
            for (int v = 0; v < NV; v++) q[j][v] = EXA_FLD(&U[p_lc * CSP + p_n0 + j * p_ns + v]);
            PDE::aux_fast(q[j], a);
#pragma unroll
            for (int v = 0; v < NV; v++) F[j][v] = 0.0;
            if (p_d == 0) PDE::template flux<0>(q[j], a, F[j]);
            else PDE::template flux<1>(q[j], a, F[j]);
        }
#pragma unroll
        for (int v = 0; v < NV; v++) {
            double qL = 0.0, FL = 0.0;
            qR[v] = 0.0;
            FR[v] = 0.0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                qL += o->phiL[j] * q[j][v];
                qR[v] += o->phiR[j] * q[j][v];
                FL += o->phiL[j] * F[j][v];
                FR[v] += o->phiR[j] * F[j][v];
            }
            qLv[v] = qL;
            if (fL >= 0) {                                       // this cell is the "+" side of its L face
                FT[fL * FS + (0 * NV + v) * N + p_t] = qL;
                FT[fL * FS + (1 * NV + v) * N + p_t] = FL;
            }
        }
        // eigenvalue of each trace state once, here (the face phase only takes the maximum)
        if (fL >= 0) LAM[(fL * 2 + 1) * N + p_t] = PDE::maxeig_fast(qLv, p_d);
        if (fR >= 0) LAM[(fR * 2 + 0) * N + p_t] = PDE::maxeig_fast(qR, p_d);
        if (p_int) {
            const double sc = dt * idx[p_d];
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    double sv = 0.0;
#pragma unroll
                    for (int j = 0; j < N; j++) sv += o->Kxi[i * N + j] * F[j][v];
                    vol[i][v] = sc * o->iw[i] * sv;
                }
        }
    }
    __syncthreads();                                             // every pencil has read the original u, tables complete

    // ---- C1: Rusanov flux of the lane's R face (it holds the minus side) with the face-wide max eigenvalue; F* overwrites the plus side's q slot
    // of this node (read by this lane only)
    if (p_task && fR >= 0) {
        double* ft = FT + fR * FS;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 2 * N; k++) s = fmax(s, EXA_FLD(&LAM[fR * 2 * N + k]));
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const double qp = EXA_FLD(&ft[(0 * NV + v) * N + p_t]), Fp = EXA_FLD(&ft[(1 * NV + v) * N + p_t]);
            FR[v] = 0.5 * (FR[v] + Fp) - 0.5 * s * (qp - qR[v]);
            ft[(0 * NV + v) * N + p_t] = FR[v];
        }
    }
    __syncthreads();
    // ---- C2: volume term + surface corrector, direction by direction (two pencils share every node)
    // (r4, measured equal and not kept: the y pencils -- a row of the cell, N NV contiguous doubles in HBM -- add theirs and store the row themselves,
    //  no copy-out phase: 0.141 against 0.139 ms per 512^2 launch, the 160-byte row stores of 64 lanes cost what the coalesced copy does)
    for (int d = 0; d < 2; d++) {
        if (p_int && p_d == d) {
            const EXA_AS4 DgOps<N>* o = ops_here<N>(ops_raw);
            const double sc = dt * idx[d];
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const double FsL = EXA_FLD(&FT[fL * FS + v * N + p_t]), FsR = FR[v];
#pragma unroll
                for (int i = 0; i < N; i++)
                    U[p_lc * CSP + p_n0 + i * p_ns + v] = EXA_FLD(&U[p_lc * CSP + p_n0 + i * p_ns + v]) + vol[i][v] - sc * o->iw[i] * (o->phiR[i] * FsR - o->phiL[i] * FsL);
            }
        }
        __syncthreads();
    }

    // ---- D: store the tile
    if (wide) {
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        constexpr int CP = CS / 2;
        for (int e = tid; e < TX * TY * CP; e += FU::NT) {
            const int c = e / CP, r = e - c * CP;
            const int lx = c / TY, ly = c - lx * TY;
            const long gx = tx0 + lx, gy = ty0 + ly;
            if (gx >= ncx || gy >= ncy) continue;                // partial tile at the end of the block
            reinterpret_cast<v2d_t*>(u_out + (gx * ncy + gy) * CS)[r] = reinterpret_cast<const v2d_t*>(U + FU::slot((lx + 1) * LY + ly + 1) * CSP)[r];
        }
    } else {
        for (int e = tid; e < TX * TY * CS; e += FU::NT) {
            const int c = e / CS, r = e - c * CS;
            const int lx = c / TY, ly = c - lx * TY;
            const long gx = tx0 + lx, gy = ty0 + ly;
            if (gx >= ncx || gy >= ncy) continue;
            u_out[(gx * ncy + gy) * CS + r] = U[FU::slot((lx + 1) * LY + ly + 1) * CSP + r];
        }
    }
}

}  // namespace exa
