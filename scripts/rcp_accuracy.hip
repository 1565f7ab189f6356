// Accuracy of v_rcp_f64 (+ Newton steps) and v_rsq_f64-based sqrt against the IEEE results, in ulp.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/rcp_accuracy.hip -o scripts/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    out[4 * i + 0] = r0; out[4 * i + 1] = r1; out[4 * i + 2] = r2; out[4 * i + 3] = 1.0 / v;
}
int main() {
    const int n = 1 << 22;
    double *hx = new double[n], *ho = new double[4 * n], *dx, *dout;
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = 0.05 + 20.0 * ((s >> 11) * (1.0 / 9007199254740992.0)); }
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 4 * n * 8, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double ref = ho[4 * i + 3], ulp = std::ldexp(1.0, std::ilogb(ref) - 52);
        for (int k2 = 0; k2 < 3; k2++) { double d = std::fabs(ho[4 * i + k2] - ref) / ulp; if (d > e[k2]) e[k2] = d; }
    }
    printf("max error vs IEEE 1/x over %d values in [0.05, 20]: v_rcp_f64 %.3g ulp | + 1 Newton step %.3g ulp | + 2 steps %.3g ulp\n", n, e[0], e[1], e[2]);
    return 0;
}
