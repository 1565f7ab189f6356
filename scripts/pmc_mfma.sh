#!/bin/bash
# MFMA / VALU counters of scripts/bin/mfma_tcontract (one --pmc pass, no trace domains), per dispatch.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_mfma}
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/mfma_counters.txt || true
cat $OUT/mfma_counters.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS \
  --output-format csv -d $OUT/run -- scripts/bin/mfma_tcontract 500 12 > $OUT/run.log 2>&1 || (tail -5 $OUT/run.log; exit 1)
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(os.path.join(root, "run", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = None
for d, c in rows.items():
    if names is None:
        names = sorted(c)
        print("dispatch " + " ".join("%22s" % n for n in names))
    print("%8d " % d + " ".join("%22.4g" % c.get(n, float("nan")) for n in names))
PY
