#!/bin/bash
# r15b — device timeline of two serving calls in flight (kernel + memory-copy trace of scripts/bench_rls.py hashed 262144; the pair
# phase is the last thing the script runs)
set -u
out=$PWD/gpurun_out/r15b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/tr" -o t -- python $REPO/scripts/bench_rls.py hashed 262144 > "$out/rls.json" 2> "$out/tr.err"
cd "$REPO"
python scripts/timeline_tail.py "$out/tr" 8 0 3 > "$out/timeline.txt"
wc -l "$out/timeline.txt"
find "$out" -type f -size +20M -delete
