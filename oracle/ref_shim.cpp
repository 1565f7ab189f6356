// ORACLE (test infrastructure only).  extern "C" entry points onto the
// reference's own C++-linkage symbols, so ctypes can call the compiled
// reference (oracle/_ref/libexa_ref.so).  Declarations restate
// `Unit test/Functions.h:2-4` and `Unit test/test.h:3`; the definitions come
// from the reference's sources compiled where they lie (oracle/Makefile).
void Flux(const double* __restrict__ Q, int normal, double* __restrict__ F);
double maxEigenvalue(const double* __restrict__ Q, int normal);
double max(double* a, double* b);
void time_step(double* Q, double dt);

extern "C" {
void ref_time_step(double* Q, double dt) { time_step(Q, dt); }
void ref_Flux(const double* Q, int normal, double* F) { Flux(Q, normal, F); }
double ref_maxEigenvalue(const double* Q, int normal) { return maxEigenvalue(Q, normal); }
double ref_max(double* a, double* b) { return max(a, b); }
// n back-to-back calls on n independent copies of one patch (CPU-baseline timing).
void ref_time_step_batched(double* Q, double dt, long n_patches, long stride) {
    for (long p = 0; p < n_patches; p++) time_step(Q + p * stride, dt);
}
}
