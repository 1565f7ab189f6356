#!/bin/bash
# One GPU-box session that produces the round's evidence under gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bench JSON lines of every configuration, rocprofv3 --kernel-trace --stats of the headline command, and per-configuration
#   HBM traffic of the dominant kernel (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, no trace domains; FETCH x2: gfx950).
# usage: scripts/profile_round.sh TAG
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
traffic() { # name  kernel-substring  json-extra  -- bench args
  name=$1; ksub=$2; extra=$3; shift 4
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
}
echo "== bench lines"; 
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; tail -c 600 $OUT/bench_cfg2.json; echo
python3 bench.py --gpus 1 --self-exchange --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg2_self_exchange.json 2> $OUT/bench_cfg2_self_exchange.err; tail -c 400 $OUT/bench_cfg2_self_exchange.json; echo
python3 bench.py --config cfg4 --gpus 1 --self-exchange --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg4_self_exchange.json 2> $OUT/bench_cfg4_self_exchange.err; tail -c 400 $OUT/bench_cfg4_self_exchange.json; echo
for cfg in cfg1 cfg4 fv-ref fv-grid cfg2_sympy cfg4_sympy; do python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; tail -c 300 $OUT/bench_$cfg.json; echo; done
echo "== kernel trace of the headline command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/trace_cfg2.log 2>&1
find $OUT/trace_cfg2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_cfg2.csv \;
head -8 $OUT/kernel_stats_cfg2.csv
for cfg in cfg1 cfg4 fv-ref fv-grid cfg2_sympy; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_$cfg.log 2>&1
  find $OUT/trace_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$cfg.csv \;
  head -5 $OUT/kernel_stats_$cfg.csv
done
echo "== traffic"
traffic cfg2 dg_stage_a_reg_kernel '{"kernel": "dg_stage_a_reg_kernel<6, exa::Euler, 2, false>", "cells": 128, "order": 5, "algorithmic_bytes_per_launch": 72477573120.0}' -- --steps 2 --warmup 1 --no-other-configs
traffic cfg1 dg_fused_single_kernel '{"algorithmic_bytes_per_launch": 1006632960.0}' -- --config cfg1 --steps 5 --warmup 2
traffic cfg4 dg_stage_a_m8_kernel '{"kernel": "dg_stage_a_m8_kernel<exa::Euler>", "cells": 64}' -- --config cfg4 --steps 2 --warmup 1
traffic fv_ref fv_rusanov_kernel '{"algorithmic_bytes_per_launch": 3690987520.0}' -- --config fv-ref --steps 5 --warmup 2
# the grid step on halo-less arrays (r4): every state read once, written once = 16 V B per volume
traffic fv_grid_4x4 'exa::FvShape<4, 1, 5, 10>, true, true>' '{"algorithmic_bytes_per_launch": 2684354560.0, "note": "2^20 patches 4x4, 10 variables: 16 x 10 B x 16 volumes"}' -- --config fv-grid --steps 10 --warmup 3
traffic fv_grid_15 'fv_rusanov_slab_kernel<exa::Euler, 1, true, false, true>' '{"algorithmic_bytes_per_launch": 2211840000.0, "note": "8192 patches 15^3, 5 variables"}' -- --config fv-grid --steps 10 --warmup 3
# the limiter's FV patch kernel (15^3 patches) inside cfg4
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for f in glob.glob(os.path.join(root, "pmc_cfg4", c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fv_rusanov_slab_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                xs.append(float(r["Counter_Value"]))
    print("fv_rusanov_slab_kernel in cfg4:", c, "n=%d mean=%.6g KiB" % (len(xs), sum(xs) / max(len(xs), 1)))
PY
echo "== matrix-instruction counters of the p = 7 stage A (32^3 cells)"
scripts/pmc_m8.sh $OUT > $OUT/pmc_m8.txt 2>&1 || tail -5 $OUT/pmc_m8.txt
tail -1 $OUT/pmc_m8.txt
echo "== SQ counters of stage A (48^3 cells)"
scripts/pmc_stage_a.sh ${TAG}sq 48 5 > $OUT/pmc_stage_a_48cubed.txt 2>&1 || tail -5 $OUT/pmc_stage_a_48cubed.txt
python3 scripts/stage_a_pmc_json.py "$OUT/pmc_stage_a_48cubed.txt" "$OUT/stage_a_pmc.json" 48 5
rm -rf $OUT/trace_*/ $OUT/pmc_*/          # keep the summaries, not the raw traces (size)
ls $OUT
