#!/bin/bash
# SQ counters of one bench configuration (two --pmc passes, no trace domains).  usage: scripts/pmc_generic.sh TAG -- <bench.py arguments>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/$n.log 2>&1 || (tail -5 $OUT/$n.log; exit 1); }
ARGS="$*"
# every library the configuration needs (incl. the generated term sets of cfg2_sympy / cfg4_sympy) built UNPROFILED first: a compiler child under --pmc is a GPU process
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
python3 scripts/pmc_summary.py $OUT
rm -rf $OUT/sq1 $OUT/sq2
