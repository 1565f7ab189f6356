set -e
L=gpurun_out/r5z_m8.log; : > $L
for t in xi xe0 xe1 xe0 xe1; do echo "== $t" >> $L; timeout -k 10 120 scripts/bin/bench_reg_n8_$t 64 2 2>&1 | grep -E "reg|max" >> $L; done
cat $L
