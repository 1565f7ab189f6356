// Probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4) on gfx950, with exact integer data.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_n8_probe.hip -o scripts/bin/mfma_n8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
    double ha[64], hb[64], hd[64], *a, *b, *d;
    for (int l = 0; l < 64; l++) { ha[l] = 1 + l; hb[l] = 100 + 3 * l; }
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
    // candidate maps: A lane = blk*16 + (ia ? i + 4k : k + 4i); B lane = blk*16 + (ib ? j + 4k : k + 4j); D lane = blk*16 + (id ? j + 4i : i + 4j)
    for (int ia = 0; ia < 2; ia++) for (int ib = 0; ib < 2; ib++) for (int id = 0; id < 2; id++) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; blk++) for (int i = 0; i < 4 && ok; i++) for (int j = 0; j < 4 && ok; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += ha[blk * 16 + (ia ? i + 4 * k : k + 4 * i)] * hb[blk * 16 + (ib ? j + 4 * k : k + 4 * j)];
            if (s != hd[blk * 16 + (id ? j + 4 * i : i + 4 * j)]) ok = false;
        }
        if (ok) printf("layout: A[i][k] in lane blk*16 + %s, B[k][j] in lane blk*16 + %s, D[i][j] in lane blk*16 + %s\n",
                       ia ? "i + 4k" : "k + 4i", ib ? "j + 4k" : "k + 4j", id ? "j + 4i" : "i + 4j");
    }
    printf("d[0..7] = %g %g %g %g %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7]);
    return 0;
}
