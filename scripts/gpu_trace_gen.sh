#!/bin/bash
# Kernel trace only (no counters) of scripts/bench_match.py -> gpurun_out/prof_<tag>/gen; prints the per-call kernel table.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_trace_gen.sh r06c'
set -u
tag=${1:-r06x}
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
REPO=$PWD
timeout 300 python scripts/bench_match.py --steps 20 ${GEN_ARGS:-} > "$out/gen_bench.json" 2> "$out/gen.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/gen" -o g -- python $REPO/scripts/bench_match.py --steps 6 ${GEN_ARGS:-} > "$out/gen_bench_under_trace.json" 2>> "$out/gen.err"
cd "$REPO"
find "$out" -type f -size +12M -delete
python scripts/summarize_gen_prof.py "$out" "$tag" 2>&1 | sed -n 1,60p
