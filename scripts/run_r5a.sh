set -e
L=gpurun_out/r5n_stage_b.log; : > $L
for r in 1 2; do
python scripts/quick_bench_stage_b.py 6 128 10 2>&1 | tail -1 >> $L
EXA_SB_SYMPY=1 python scripts/quick_bench_stage_b.py 6 128 10 2>&1 | tail -1 | sed 's/^/sympy: /' >> $L
done
EXA_SB_SYMPY=1 python scripts/quick_bench_stage_b.py 8 64 10 2>&1 | tail -1 | sed 's/^/sympy: /' >> $L
python scripts/quick_bench_stage_b.py 8 64 10 2>&1 | tail -1 >> $L
cat $L
python -m pytest tests/test_user_pde.py -m gpu -x -q 2>&1 | tail -2
