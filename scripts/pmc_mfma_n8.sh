#!/bin/bash
# Matrix-pipe counters (own pass, no trace domains) of scripts/bin/mfma_n8 and of the N = 8 stage A in scripts/bin/bench_reg_n8.  usage: scripts/pmc_mfma_n8.sh OUTTAG
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/micro -- scripts/bin/mfma_n8 500 > $OUT/micro.log 2>&1 || (tail -5 $OUT/micro.log; exit 1)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/kern -- scripts/bin/bench_reg_n8 32 1 > $OUT/kern.log 2>&1 || (tail -5 $OUT/kern.log; exit 1)
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for sub in ("micro", "kern"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "step_ops" in k: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(sub, k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
rm -rf $OUT/micro $OUT/kern
