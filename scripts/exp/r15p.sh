#!/bin/bash
# r15p — where the engine's mutex-held part of a serving call goes with calls in flight ([wm] laps, experiment build): k_resp<true>
# into the staging (0) against copy commands (1)
set -u
out=$PWD/gpurun_out/r15p; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp RL_WIRE_TRACE=1
for v in 0 1; do
  RL_RESP_VIA_COPY=$v timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$v.json" 2> "$out/laps_$v.txt"
  echo "via_copy=$v: $(python -c "import json;d=json.load(open('$out/rls_$v.json'))['sizes']['262144'];print(round(d['with_headers']['p50_ms'],3), round(d['with_headers_two_in_flight']['ms_per_batch_sustained'],3))")"
  grep "\[wm\]" "$out/laps_$v.txt" | tail -n 60 > "$out/tail_$v.txt"
done
