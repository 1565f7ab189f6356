#!/bin/bash
# Matrix-instruction counters of the LIBRARY's p = 7 stage A (dg_stage_a_m8_kernel) under bench.py --config cfg4 at 32^3 cells: one --pmc pass, no trace
# domains.  Writes $OUT/m8_pmc.json (copy to profiles/m8_pmc.json: bench.py reports it as roofline.mfma_busy).  usage: scripts/pmc_m8.sh OUTDIR
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_m8 -- python3 bench.py --config cfg4 --cells 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_m8.log 2>&1 || (tail -5 $OUT/pmc_m8.log; exit 1)
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(list)
name = None
for f in glob.glob(os.path.join(root, "pmc_m8", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dg_stage_a_m8_kernel" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("exa::dg", "dg")
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
# SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4), each with 32 SIMDs: SIMD time of the launch = 32 x SQ_BUSY_CYCLES
simd = 32.0 * m["SQ_BUSY_CYCLES"]
rec = {"kernel": name, "cells": 32, "launches_averaged": len(agg["SQ_INSTS_MFMA"]), "counters": m,
       "mfma_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, "valu_issue_frac": 4.0 * m["SQ_INSTS_VALU"] / simd,
       "mfma_cycles_per_instruction": m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"],
       "source": "scripts/pmc_m8.sh: rocprofv3 --pmc SQ_* on bench.py --config cfg4 --cells 32; SIMD time = 32 x SQ_BUSY_CYCLES"}
json.dump(rec, open(os.path.join(root, "m8_pmc.json"), "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/pmc_m8
