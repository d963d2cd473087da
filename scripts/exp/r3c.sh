#!/bin/bash
set -u
out=$PWD/gpurun_out/r3c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_d1.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --depth 1 --timing-mode 0 > "$out/d1.json" 2> "$out/d1.err"
python scripts/apply_trace.py $out/trace_d1.bin
RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_d3.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 > "$out/d3.json" 2> "$out/d3.err"
python scripts/apply_trace.py $out/trace_d3.bin
