#!/bin/bash
# r5d: rocprofv3 kernel trace + stats of the bench command at the round's last commit (timing only, no counters).
set -u
out=$PWD/gpurun_out/prof_r03d; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0"
cd /tmp
timeout 22 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $BENCH > "$out/bench_under_trace.json" 2> "$out/trace.err"
cd "$OLDPWD"
find "$out" -type f -size +8M -delete
ls -R "$out" | head -20
