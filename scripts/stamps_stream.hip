// Diagnostic: per-phase cycle shares of dg_stage_a_stream_kernel<N,Euler> (first wave of each 256-lane group).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DEXA_STAMPS [-DSN=7] -I exahype_amd/csrc scripts/stamps_stream.hip -o scripts/stamps_stream
#include <cstdio>
#include <vector>
#include "exa_dg_stream.hpp"
using namespace exa;
#ifndef SN
#define SN 8
#endif
#ifndef SHS
#define SHS 2
#define SOH 2
#endif
int main() {
    constexpr int N = SN;
    using SA = StageAStream<N, Euler, SHS, SOH>;
    const long nc = 20, ncells = nc * nc * nc, NN = N * N * N;
    const long ndof = ncells * NN * 5, ntr = 3 * 2 * ncells * 2 * 5 * N * N;
    std::vector<double> h(ndof);
    for (long i = 0; i < ndof; i++) { int v = i % 5; h[i] = v == 0 ? 1.0 + 0.1 * ((i * 7919) % 100) / 100.0 : (v == 4 ? 2.5 + 0.1 * ((i * 104729) % 100) / 100.0 : 0.1 * (((i * 31) % 100) / 100.0 - 0.5)); }
    double *u, *tr, *slab; void* ops;
    const int grid = 256;
    hipMalloc(&u, ndof * 8); hipMalloc(&tr, ntr * 8); hipMalloc(&ops, sizeof(DgOps<N>)); hipMalloc(&slab, grid * SA::SLAB_D * 8);
    hipMemcpy(u, h.data(), ndof * 8, hipMemcpyHostToDevice);
    DgOps<N> o{};
    for (int i = 0; i < N; i++) { o.w[i] = 1.0 / N; o.iw[i] = N; o.phiL[i] = 0.1; o.phiR[i] = 0.1; o.Tsum[i] = 0.06; for (int j = 0; j < N; j++) { o.D[i*N+j] = 0.01*(i-j); o.DT[j*N+i] = 0.01*(i-j); o.Kxi[i*N+j] = 0.01; o.T[i*N+j] = 0.01; } }
    hipMemcpy(ops, &o, sizeof(o), hipMemcpyHostToDevice);
    auto kern = dg_stage_a_stream_kernel<N, Euler, SHS, SOH>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SA::LDS_BYTES);
    CellBox box; for (int d = 0; d < 3; d++) { box.nc[d] = nc; box.lo[d] = 0; box.nb[d] = nc; } box.nbox = ncells;
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long z[48] = {0};
#ifdef EXA_STAMPS
        hipMemcpyToSymbol(HIP_SYMBOL(g_exa_stamps), z, sizeof(z));
#endif
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SA::NT), SA::LDS_BYTES, 0, u, u, tr, ncells, box, 1e-5, nc * 1.0, nc * 1.0, nc * 1.0, N, ops, slab);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef EXA_STAMPS
        hipMemcpyFromSymbol(z, HIP_SYMBOL(g_exa_stamps), sizeof(z));
#endif
        const char* names[12] = {"load", "load barrier", "D work", "D barrier", "x store", "x barrier", "fold", "fold barrier", "new q + fence", "averages (F1)", "vol+traces (F2)", "store (F3)"};
        if (rep == 0) continue;
        printf("N=%d HS=%d OH=%d, %ld cells, %.2f ms; cycles per cell, first wave of each 256-lane group (x, y, z, owners-only):\n", N, SHS, SOH, ncells, ms);
        for (int k = 0; k < 12; k++) printf("   %-18s %9.0f %9.0f %9.0f %9.0f\n", names[k], (double)z[k] / ncells, (double)z[12 + k] / ncells, (double)z[24 + k] / ncells, (double)z[36 + k] / ncells);
        double tot = 0; for (int k = 0; k < 12; k++) tot += z[k];
        printf("   total              %9.0f\n", tot / ncells);
    }
    return 0;
}
