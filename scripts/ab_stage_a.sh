#!/bin/bash
# Development aid: stage-A launch time of several library variants (exahype_amd.build --variant), same GPU box, two rounds.
# usage: scripts/ab_stage_a.sh N cells tag1 tag2 ...
N=$1; C=$2; shift 2
for round in 1 2; do
  for t in "$@"; do
    printf "%-14s " $t
    EXA_LIB=exahype_amd/lib/var_$t/libexahype_hip.so python scripts/quick_bench_stage_a.py $N $C 5 2>&1 | grep "stage A"
  done
done
