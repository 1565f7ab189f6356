#!/bin/bash
# HBM traffic of one stage-A launch of the bench workload (FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes, no trace domains).  Writes gpurun_out/stage_a_traffic.json; copy it to profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CELLS=${1:-128}
OUT=gpurun_out/pmc_traffic_$CELLS
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 2 --warmup 1 --cells $CELLS --no-cpu-baseline > $OUT/$c.log 2>&1 || (tail -5 $OUT/$c.log; exit 1)
done
python3 - "$OUT" "$CELLS" <<'PY'
import csv, glob, json, os, sys
root, cells = sys.argv[1], int(sys.argv[2])
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "dg_stage_a_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                xs.append(float(r["Counter_Value"]))
    val[c] = sum(xs) / len(xs)
# rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read stream
# (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE is exact for wide streaming stores.
fetch = 2 * val["FETCH_SIZE"] * 1024
write = val["WRITE_SIZE"] * 1024
rec = {"cells": cells, "order": 5, "kernel": "dg_stage_a_kernel<3,6,Euler>", "fetch_bytes_corrected": fetch,
       "write_bytes": write, "hbm_bytes_per_launch": fetch + write, "raw_kib": val,
       "algorithmic_bytes_per_launch": cells ** 3 * 34560.0}
json.dump(rec, open("gpurun_out/stage_a_traffic.json", "w"), indent=1)
print(json.dumps(rec))
PY
