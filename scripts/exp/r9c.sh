#!/bin/bash
# r9c — the in-place compaction rebuilt as a re-insertion (r9a: the max(home, cursor) packing lost cells): GPU suite, sweep bench + trace.
set -u
out=$PWD/gpurun_out/r9c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$out/pytest_gpu.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"
timeout 300 python scripts/bench_sweep.py > "$out/sweep.json" 2> "$out/sweep.err"
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/sweep_trace" -o t -- python $OLDPWD/scripts/bench_sweep.py --reps 5 > "$out/sweep_under_trace.json" 2> "$out/sweep_trace.err" )
find "$out" -type f -size +4M -delete
tail -n 5 "$out/pytest_gpu.log"; cat "$out/sweep.json"; grep -h "k_compact\|k_scan<3>" "$out"/sweep_trace/*kernel_stats.csv | cut -c1-60,150-260
timeout 120 python scripts/bench_per_request.py > "$out/per_request.json" 2> "$out/per_request.err"; tail -c 600 "$out/per_request.json"
