#!/bin/bash
# PMC passes (SQ counters, own runs, no trace domains) of an FV timing script on the GPU box.  usage: scripts/pmc_fv.sh OUTTAG script.py [args]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
run() { n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $SCRIPT > $OUT/$n.log 2>&1 || (tail -5 $OUT/$n.log; exit 1)
}
SCRIPT="$*"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
run sq3 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
python3 scripts/pmc_summary.py $OUT
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
