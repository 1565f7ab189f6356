set -e
L=gpurun_out/r5r_defer.log; : > $L
V=exahype_amd/lib/var_defer2/libexahype_hip.so
for r in 1 2; do
python scripts/quick_bench_stage_a.py 6 128 3 2>&1 | tail -1 | sed 's/^/euler default: /' >> $L
EXA_LIB=$V python scripts/quick_bench_stage_a.py 6 128 3 2>&1 | tail -1 | sed 's/^/euler defer2: /' >> $L
done
python scripts/quick_bench_sympy.py 6 64 3 2>&1 | tail -3 | sed 's/^/sympy default: /' >> $L
EXA_EXTRA_FLAGS="-DEXA_REG_DEFER_FOLD=2" python scripts/quick_bench_sympy.py 6 64 3 2>&1 | tail -3 | sed 's/^/sympy defer2: /' >> $L
python scripts/quick_bench_plain.py 6 32 xt 2>&1 | tail -1 | sed 's/^/xt default: /' >> $L
EXA_EXTRA_FLAGS="-DEXA_REG_DEFER_FOLD=2" python scripts/quick_bench_plain.py 6 32 xt 2>&1 | tail -1 | sed 's/^/xt defer2: /' >> $L
cat $L
