#!/bin/bash
set -u
out=$PWD/gpurun_out/r3j; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_d2.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 --depth 2 > "$out/d2.json" 2> "$out/d2.err"
python scripts/apply_trace.py $out/trace_d2.bin | head -12
RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_u2.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 --zipf 0 --depth 2 > "$out/u2.json" 2> "$out/u2.err"
python scripts/apply_trace.py $out/trace_u2.bin | head -12
