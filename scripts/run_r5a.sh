set -e
L=gpurun_out/r5x_m8.log; : > $L
for t in cur pu cur pu; do echo "== $t" >> $L; timeout -k 10 120 scripts/bin/bench_reg_n8_$t 64 2 2>&1 | grep -E "reg|max" >> $L; done
cat $L
