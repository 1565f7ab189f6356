// A compiled C++ host on the C-ABI of include/exahype_hip.h, running the protocol of the reference's own test driver
// (`Unit test/correctness_test.cpp`): the sin input of initInputData (:102-106), the sizes of main (:177-183), ONE call where the reference calls
// `time_step(Q1, 1)` (:195), and the reference's exact `!=` comparison (:199-204) -- here against the golden values the compiled reference produced for
// that input (tests/golden/fv_ref2d_sin.json, handed over by the test as "index hex-double" lines: the reference's second implementation,
// `old_time_step`, needs Peano headers the image does not hold).  Plain C++17, no HIP header, no Python: built by g++ against the header and
// -lexahype_hip (tests/test_compiled_host.py).  usage: correctness_host <golden.txt>      exit code 0 = every defined output equal bit for bit
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "exahype_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <golden.txt>\n", argv[0]); return 2; }
    const int dim = 2, patch_size = 4, halo_size = 1, n_real = 5, n_aux = 5;
    int no_inputs = n_real + n_aux;
    for (int i = 0; i < dim; i++) no_inputs *= patch_size + 2 * halo_size;

    std::vector<double> Q1(no_inputs), Q0(no_inputs);
    for (int i = 0; i < no_inputs; i++) Q0[i] = Q1[i] = std::sin(3.141 * i / no_inputs);

    exa_fv_plan* plan = nullptr;
    if (exa_fv_plan_create(/*device*/ 0, EXA_FV_FAITHFUL, dim, patch_size, halo_size, n_real, n_aux, /*n_patches*/ 1, EXA_PDE_EULER_REF2D, &plan) != EXA_OK) {
        std::fprintf(stderr, "exa_fv_plan_create: %s\n", exa_last_error());
        return 3;
    }
    if (exa_fv_q_count(plan) != no_inputs) { std::fprintf(stderr, "exa_fv_q_count = %ld, expected %d\n", exa_fv_q_count(plan), no_inputs); return 3; }
    if (exa_fv_time_step_host(plan, Q1.data(), /*dt*/ 1.0, /*h*/ 1.0) != EXA_OK) {          // replaces: time_step(Q1, 1);
        std::fprintf(stderr, "exa_fv_time_step_host: %s\n", exa_last_error());
        return 3;
    }
    exa_fv_plan_destroy(plan);

    // golden file: "v <index> <hex double>" = an output the reference defines; "p <index>" = an entry the reference leaves as it was
    std::FILE* f = std::fopen(argv[1], "r");
    if (!f) { std::perror(argv[1]); return 2; }
    int bad = 0, seen = 0, idx;
    char kind, hex[64];
    while (std::fscanf(f, " %c %d", &kind, &idx) == 2) {
        if (idx < 0 || idx >= no_inputs) { std::fprintf(stderr, "golden index %d out of range\n", idx); return 2; }
        double want = Q0[idx];
        if (kind == 'v') {
            if (std::fscanf(f, " %63s", hex) != 1) return 2;
            want = std::strtod(hex, nullptr);
        }
        seen++;
        if (Q1[idx] != want) {                                    // the reference's comparison: exact
            if (bad < 8) std::fprintf(stderr, "Q[%d] = %a, reference %a\n", idx, Q1[idx], want);
            bad++;
        }
    }
    std::fclose(f);
    if (seen == 0) { std::fprintf(stderr, "empty golden file\n"); return 2; }
    if (bad > 0) {
        std::printf("%d of %d checked entries differ from the reference\n", bad, seen);
        return 1;
    }
    std::printf("correct: %d entries equal the reference bit for bit\n", seen);
    return 0;
}
