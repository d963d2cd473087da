#!/bin/bash
# r9g — the general resolver's bucket count (RL_GEN_BUCKET_LOG2, experiment build): fewer, longer buckets = longer runs for k_bkt_scatter's stores.
set -u
out=$PWD/gpurun_out/r9g; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for b in 11 10 9 8; do
  RL_GEN_BUCKET_LOG2=$b timeout 200 python scripts/bench_match.py --steps 20 2> "$out/b$b.err" | cut -c1-330 > "$out/b$b.json"; echo "$b $(cut -c150-330 "$out/b$b.json")"
done
