#!/bin/bash
# r15g — host laps ([rli], experiment build) of four serving calls in flight, responses through copy commands or not
set -u
out=$PWD/gpurun_out/r15g; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp RLI_TRACE=1
ulimit -c 0
for v in 0 1; do
  RL_RESP_VIA_COPY=$v timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$v.json" 2> "$out/laps_$v.txt"
  grep "\[rli\]" "$out/laps_$v.txt" | tail -n 120 > "$out/laps_tail_$v.txt"
  wc -l "$out/laps_tail_$v.txt"
done
