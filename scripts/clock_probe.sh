#!/bin/bash
# Engine clock while the stage-A kernel of cfg 2 runs (development aid): samples rocm-smi beside a 64^3 launch loop.
cd "$GRAFT_REPO_ROOT"
python3 scripts/quick_bench_stage_a.py 6 128 200 > /tmp/qa.log 2>&1 &
PID=$!
sleep 18
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -4; sleep 0.4; done
wait $PID
tail -1 /tmp/qa.log
echo idle:; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
