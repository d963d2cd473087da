#!/bin/bash
# r15e — r15b's trace (four serving calls in flight) with the response bytes leaving in copy commands (experiment build, RL_RESP_VIA_COPY=1)
set -u
out=$PWD/gpurun_out/r15e; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp RL_RESP_VIA_COPY=${VIA:-1}
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/tr" -o t -- python $REPO/scripts/bench_rls.py hashed 262144 > "$out/rls.json" 2> "$out/tr.err"
cd "$REPO"
python scripts/timeline_tail.py "$out/tr" 8 0 3 > "$out/timeline.txt"
wc -l "$out/timeline.txt"
find "$out" -type f -size +20M -delete
