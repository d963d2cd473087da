#!/bin/bash
# Kernel + memory-copy trace of scripts/bench_sharded_requests.py (world 1) -> gpurun_out/prof_<tag>/sh; prints one step's timeline.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_trace_sharded.sh r06y'
set -u
tag=${1:-r06y}
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
REPO=$PWD
timeout 300 python scripts/bench_sharded_requests.py > "$out/sharded.json" 2> "$out/sh.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/sh" -o s -- python $REPO/scripts/bench_sharded_requests.py 262144 6 > "$out/sharded_under_trace.json" 2>> "$out/sh.err"
cd "$REPO"
find "$out" -type f -size +12M -delete
python scripts/timeline_step.py "$out/sh" k_route_count 2>&1 | sed -n 1,120p
