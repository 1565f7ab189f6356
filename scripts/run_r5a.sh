set -e
mkdir -p gpurun_out
L=gpurun_out/r5h_m8.log
: > $L
for t in xi pl pl_noilp pl_s0 xi; do echo "== $t" >> $L; timeout -k 10 120 scripts/bin/bench_reg_n8_$t 64 2 2>&1 | grep -E "reg|max" >> $L; done
echo "== stamps pl" >> $L; timeout -k 10 120 scripts/bin/bench_reg_n8_pl_stamps 32 2 >> $L 2>&1
cat $L
