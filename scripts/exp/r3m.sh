#!/bin/bash
set -u
out=$PWD/gpurun_out/r3m; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
RL_APPLY_TRACE=1 timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 --depth 2 > "$out/d2.json" 2> "$out/d2.err"
grep "^\[part\]\|^\[apply\]" $out/d2.err | tail -6
RL_APPLY_TRACE=1 timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 --depth 2 --zipf 0 > "$out/u2.json" 2> "$out/u2.err"
grep "^\[part\]\|^\[apply\]" $out/u2.err | tail -4
