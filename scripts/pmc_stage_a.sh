#!/bin/bash
# PMC passes for the stage-A kernel (run on the GPU box from the repo root).
# Counters are collected in their own runs (no trace domains), per the gpurun rules.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
CELLS=${2:-48}
ORDER=${3:-5}
mkdir -p $OUT
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 bench.py --steps 2 --warmup 1 --cells $CELLS --order $ORDER --no-cpu-baseline --no-other-configs > $OUT/$n.log 2>&1 || (tail -5 $OUT/$n.log; exit 1)
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT
