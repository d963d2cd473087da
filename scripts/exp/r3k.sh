#!/bin/bash
set -u
out=$PWD/gpurun_out/r3k; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for x in 0 1 2 4 7; do
RL_PART_EXP=$x RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_$x.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 --depth 2 > "$out/d_$x.json" 2> "$out/d_$x.err"
echo "== exp $x"; python scripts/apply_trace.py $out/trace_$x.bin | sed -n 2,8p
done
