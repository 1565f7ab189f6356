#!/bin/bash
# HBM traffic of one kernel of a bench configuration: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (no trace domains), FETCH x2 (gfx950:
# MI355X_MICROARCH.md, HBM section).  Writes $OUT/traffic_NAME.json.      usage: scripts/pmc_traffic_generic.sh OUTDIR NAME KERNEL-SUBSTRING JSON-EXTRA -- <bench.py arguments>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; name=$2; ksub=$3; extra=$4; shift 5
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1        # (no compiler child under the profiler's preload)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$name/$c -- python3 bench.py "$@" --no-cpu-baseline > $OUT/pmc_$name.$c.log 2>&1 || (tail -5 $OUT/pmc_$name.$c.log; exit 1)
done
python3 - "$OUT/pmc_$name" "$ksub" "$OUT/traffic_$name.json" "$extra" <<'PY'
import csv, glob, json, os, sys
root, ksub, out, extra = sys.argv[1:5]
val, n = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ksub in r["Kernel_Name"] and r["Counter_Name"] == c:
                xs.append(float(r["Counter_Value"]))
    val[c], n[c] = sum(xs) / max(len(xs), 1), len(xs)
rec = {"kernel": ksub, "launches_averaged": n, "raw_kib": val, "fetch_bytes_corrected": 2 * val["FETCH_SIZE"] * 1024,
       "write_bytes": val["WRITE_SIZE"] * 1024, "hbm_bytes_per_launch": 2 * val["FETCH_SIZE"] * 1024 + val["WRITE_SIZE"] * 1024}
rec.update(json.loads(extra))
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/pmc_$name
