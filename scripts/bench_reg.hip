// Development aid: stage A at N = 6 (3-D Euler) outside Python -- the register-resident kernel (exa_dg_reg.hpp) and the LDS-resident
// one (exa_dg_kernels.hpp) timed in interleaved rounds in ONE process on the same random data, with the library's own operator
// image and lane tables; with -DEXA_STAMPS the per-phase cycle stamps of the register-resident kernel (one column per wave).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed [-DEXA_STAMPS] -I exahype_amd/csrc scripts/bench_reg.hip -o scripts/bin/bench_reg
// Run:   scripts/bin/bench_reg [cells per axis = 64] [rounds = 3]
#define EXA_DIM 3
#define EXA_PDE_ID 1
#include <cstdarg>
#include <cstdlib>
#include <random>
#include "dg_inst.hip"
#include "dg_operators_host.cpp"
namespace exa { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
using namespace exa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
#ifndef BN
#define BN 6                                  // -DBN=8: the N = 8 kernels (exa_dg_m8.hpp against exa_dg_stream.hpp)
#endif
    constexpr int N = BN, NN3 = N * N * N, NF2 = N * N;
    const long nc = argc > 1 ? atol(argv[1]) : 64, ncells = nc * nc * nc;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const long ndof = ncells * NN3 * 5, ntr = 3 * 2 * ncells * 2 * 5 * NF2;
    std::vector<double> h(ndof);
    std::mt19937_64 rng(4);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (long i = 0; i < ndof; i += 5) { const double rho = 1 + 0.2 * U(rng); h[i] = rho; for (int a = 1; a < 4; a++) h[i + a] = rho * (0.4 * U(rng) - 0.2); h[i + 4] = 2.6 + 0.5 * U(rng); }
    double *u0, *u, *tr;
    CK(hipMalloc(&u0, ndof * 8)); CK(hipMalloc(&u, ndof * 8)); CK(hipMalloc(&tr, ntr * 8));
    CK(hipMemcpy(u0, h.data(), ndof * 8, hipMemcpyHostToDevice));
    DgOpsHost ops{};
    if (build_dg_operators(N, &ops)) return 1;
    const size_t ob = ops_image(N, &ops, nullptr);
    std::vector<char> img(ob);
    ops_image(N, &ops, img.data());
    CK(hipMalloc(&ops.dev, ob)); CK(hipMemcpy(ops.dev, img.data(), ob, hipMemcpyHostToDevice));
    CellBox box; for (int d = 0; d < 3; d++) { box.nc[d] = nc; box.lo[d] = 0; box.nb[d] = nc; } box.nbox = ncells;
    const double dx = 1.0 / nc, idx[3] = {1 / dx, 1 / dx, 1 / dx}, dt = 0.05 * dx / (2 * N - 1) / 3 / 2.5;
    const double Nd = NN3, flop = (N * (2.0 * 4 * 5 * Nd * N * N + 3 * 20.0 * Nd * N) + 2.0 * 4 * 5 * Nd * N + 2.0 * 3 * 5 * Nd * N + 8.0 * 3 * 5 * Nd) * ncells;   // exa_dg_work()
    if (N > 6) { ops.scratch = nullptr; CK(hipMalloc(&ops.scratch, scratch_bytes(N))); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> ref;
    for (int r = 0; r < rounds + 1; r++)
        for (int variant = 2; variant >= 1; variant--) {
            ops.stage_a_variant = variant;
            CK(hipMemcpy(u, u0, ndof * 8, hipMemcpyDeviceToDevice));
#ifdef EXA_STAMPS
            unsigned long long z[48] = {0};
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_exa_stamps), z, sizeof(z)));
#endif
            CK(hipEventRecord(e0));
            if (stage_a(N, u, u, tr, ncells, &box, dt, idx, N, &ops, 0)) return 1;
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 0) {                                                // first round: warm-up and cross-check of the two kernels
                std::vector<double> out(NN3 * 5 * 64);
                CK(hipMemcpy(out.data(), u + (ncells / 2) * NN3 * 5, out.size() * 8, hipMemcpyDeviceToHost));
                if (variant == 2) ref = out;
                else { double e = 0; for (size_t i = 0; i < out.size(); i++) e = std::max(e, std::abs(out[i] - ref[i])); printf("max |reg - lds| over 64 cells: %.3e\n", e); }
                continue;
            }
            printf("%s  %8.3f ms  %6.2f TFLOP/s  (%.4f of 78.6)\n", variant == 2 ? "reg" : "lds", ms, flop / ms * 1e-9, flop / ms * 1e-9 / 78.6);
#ifdef EXA_STAMPS
            if (variant == 2 && r == rounds) {
                CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_exa_stamps), sizeof(z)));
#if BN == 8
                const char* names[12] = {"it0 load", "it0 barrier", "it0 derive", "it0 fold", "(fold tail)", "barrier", "derive", "fold+load", "averages", "vol+traces", "u* store", "-"};
#else
                const char* names[12] = {"it0 load", "it0 barrier", "it0 derive", "it0 barrier", "it0 fold", "load", "barrier", "derive", "barrier", "fold", "averages", "vol+traces"};
#endif
                printf("cycles per cell (waves 0..3):\n");
                double tot[4] = {0};
                for (int k = 0; k < 12; k++) { printf("   %-12s", names[k]); for (int w = 0; w < 4; w++) { printf(" %9.0f", (double)z[w * 12 + k] / ncells); tot[w] += (double)z[w * 12 + k] / ncells; } printf("\n"); }
                printf("   %-12s %9.0f %9.0f %9.0f %9.0f   (u* store and the cell-loop overhead are in 'it0 load')\n", "total", tot[0], tot[1], tot[2], tot[3]);
            }
#endif
        }
    return 0;
}
