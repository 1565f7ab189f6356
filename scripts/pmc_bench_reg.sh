#!/bin/bash
# PMC passes (SQ counters, own runs, no trace domains) of scripts/bin/bench_reg[_TAG] on the GPU box.  usage: scripts/pmc_bench_reg.sh OUTTAG [BINTAG] [cells]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
BIN=scripts/bin/bench_reg${2:+_$2}
CELLS=${3:-48}
mkdir -p $OUT
run() { n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- $BIN $CELLS 1 > $OUT/$n.log 2>&1 || (tail -5 $OUT/$n.log; exit 1)
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
python3 scripts/pmc_summary.py $OUT | grep -v "step_ops" 
rm -rf $OUT/sq1 $OUT/sq2
