#!/bin/bash
# general resolver: one-pass matcher + pass-end status word
set -u
out=$PWD/gpurun_out/r4c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_match.py -m gpu -q -x 2>&1 | tail -2
for v in "X=1" "RL_MATCH_ONE=0"; do
  echo "== $v"; env $v timeout 200 python scripts/bench_match.py --steps 20 2>&1 | tail -1 | cut -c1-400
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python scripts/bench_match.py --steps 6 > $out/match_prof.json 2> $out/trace.err
find $out -type f -size +6M -delete
