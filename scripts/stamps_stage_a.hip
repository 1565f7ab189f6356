// Diagnostic: per-phase cycle shares of dg_stage_a_kernel<3,6,Euler> (wave 0 of every workgroup).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DEXA_STAMPS -I exahype_amd/csrc scripts/stamps_stage_a.hip -o scripts/stamps_stage_a
#include <cstdio>
#include <vector>
#include <algorithm>
#include "exa_dg_kernels.hpp"
using namespace exa;
int main() {
    constexpr int N = 6, DIM = 3, CPB = 1;
    using SA = StageA<DIM, N, Euler, CPB>;
    const long nc = 32, ncells = nc * nc * nc;
    const long ndof = ncells * 216 * 5, ntr = 3 * 2 * ncells * 2 * 5 * 36;
    std::vector<double> h(ndof);
    for (long i = 0; i < ndof; i++) { int v = i % 5; h[i] = v == 0 ? 1.0 + 0.1 * ((i * 7919) % 100) / 100.0 : (v == 4 ? 2.5 + 0.1 * ((i * 104729) % 100) / 100.0 : 0.1 * (((i * 31) % 100) / 100.0 - 0.5)); }
    double *u, *tr; void* ops;
    hipMalloc(&u, ndof * 8); hipMalloc(&tr, ntr * 8); hipMalloc(&ops, SA::IMAGE_BYTES);
    hipMemcpy(u, h.data(), ndof * 8, hipMemcpyHostToDevice);
    DgOps<N> o{};   // the values do not matter for timing shares; keep them finite and small
    for (int i = 0; i < N; i++) { o.w[i] = 1.0 / N; o.iw[i] = N; o.phiL[i] = 0.1; o.phiR[i] = 0.1; o.Tsum[i] = 0.06; for (int j = 0; j < N; j++) { o.D[i*N+j] = 0.01*(i-j); o.DT[j*N+i] = 0.01*(i-j); o.Kxi[i*N+j] = 0.01; o.T[i*N+j] = 0.01; } }
    hipMemcpy(ops, &o, sizeof(o), hipMemcpyHostToDevice);
    {   // natural task enumeration (the library builds a conflict-free one: dg_inst.hip OpsImage)
        std::vector<int> perm(DIM * SA::WPD * 64);
        for (int d = 0; d < DIM; d++) for (int k = 0; k < SA::WPD * 64; k++) perm[d * SA::WPD * 64 + k] = k < SA::TD ? k : -1;
        hipMemcpy((char*)ops + SA::PERM_OFF, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice);
        DgStepOps<N> so;
        for (int k = 0; k < N * N; k++) so.Tdt[k] = -1e-7;
        for (int k = 0; k < N; k++) so.Tsdt[k] = -6e-7;
        hipMemcpy((char*)ops + SA::STEP_OFF, &so, sizeof(so), hipMemcpyHostToDevice);
    }
    auto kern = dg_stage_a_kernel<DIM, N, Euler, CPB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SA::LDS_BYTES);
    CellBox box; for (int d = 0; d < 3; d++) { box.nc[d] = nc; box.lo[d] = 0; box.nb[d] = nc; } box.nbox = ncells;
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long z[48] = {0};
#ifdef EXA_STAMPS
        hipMemcpyToSymbol(HIP_SYMBOL(g_exa_stamps), z, sizeof(z));
#endif
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(SA::NT), SA::LDS_BYTES, 0, u, u, tr, ncells, box, 1e-5, nc * 1.0, nc * 1.0, nc * 1.0, N, ops);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) printf("launch %.3f ms\n", ms);
#ifdef EXA_STAMPS
        hipMemcpyFromSymbol(z, HIP_SYMBOL(g_exa_stamps), sizeof(z));
#endif
        const char* names[12] = {"load+init", "D work", "D barrier", "T work", "T barrier", "T: loads done", "-", "avg (F1)", "vol+traces (F2)", "store (F3)", "z store", "z barrier"};
        if (rep == 0) continue;
        printf("cycles per cell, first wave of each direction group (x, y, z):\n");
        for (int k = 0; k < 12; k++) printf("   %-18s %9.0f %9.0f %9.0f\n", names[k], (double)z[k] / ncells, (double)z[12 + k] / ncells, (double)z[24 + k] / ncells);
        double tot = 0; for (int k = 0; k < 12; k++) tot += z[k];
        printf("   total              %9.0f\n", tot / ncells);
#ifdef EXA_STAMPS
        {   // load balance of the persistent grid: per-workgroup start / end on the constant-rate clock (100 MHz ticks)
            std::vector<unsigned long long> sp(2 * 1024);
            hipMemcpyFromSymbol(sp.data(), HIP_SYMBOL(g_exa_wg_span), sp.size() * 8);
            unsigned long long t0 = ~0ull, te_min = ~0ull, te_max = 0, ts_max = 0; double dsum = 0, dmin = 1e30, dmax = 0;
            for (int b = 0; b < 256; b++) { t0 = std::min(t0, sp[2 * b]); }
            for (int b = 0; b < 256; b++) {
                const double d = (double)(sp[2 * b + 1] - sp[2 * b]);
                dsum += d; dmin = std::min(dmin, d); dmax = std::max(dmax, d);
                te_min = std::min(te_min, sp[2 * b + 1]); te_max = std::max(te_max, sp[2 * b + 1]); ts_max = std::max(ts_max, sp[2 * b]);
            }
            printf("workgroup spans (ticks of the constant clock): mean %.0f min %.0f max %.0f; last start %+lld, first end %lld, last end %lld after the first start\n",
                   dsum / 256, dmin, dmax, (long long)(ts_max - t0), (long long)(te_min - t0), (long long)(te_max - t0));
            double xs[8] = {0}; for (int b = 0; b < 256; b++) xs[b % 8] += (double)(sp[2 * b + 1] - sp[2 * b]) / 32;
            printf("mean span per XCD (blockIdx %% 8):"); for (int x = 0; x < 8; x++) printf(" %.0f", xs[x]); printf("\n");
        }
#endif
    }
    return 0;
}
