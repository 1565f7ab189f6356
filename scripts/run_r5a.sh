set -e
L=gpurun_out/r5w_n6_flags.log; : > $L
for t in f0 f3 f1 f8 f0; do echo "== $t" >> $L; timeout -k 10 200 scripts/bin/bench_reg_n6_$t 64 2 2>&1 | grep -E "reg " >> $L; done
cat $L
