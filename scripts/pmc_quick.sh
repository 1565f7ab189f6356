#!/bin/bash
# Development aid: FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains) of a stage-A launch.
# usage: scripts/pmc_quick.sh OUTDIR KERNEL_SUBSTRING -- python3 script args...   (EXA_LIB may select a library variant)
set -e
OUT=$1; KSUB=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- "$@" > $OUT/$c.log 2>&1 || (tail -5 $OUT/$c.log; exit 1)
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys
root, ksub = sys.argv[1], sys.argv[2]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ksub in r["Kernel_Name"] and r["Counter_Name"] == c:
                xs.append(float(r["Counter_Value"]))
    v = sum(xs) / max(len(xs), 1)
    print("%s %s: n=%d mean=%.6g KiB -> %.3f GB%s" % (ksub, c, len(xs), v, v * 1024 / 1e9 * (2 if c == "FETCH_SIZE" else 1), " (x2 gfx950 correction applied)" if c == "FETCH_SIZE" else ""))
PY
