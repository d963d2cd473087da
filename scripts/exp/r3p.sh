#!/bin/bash
set -u
out=$PWD/gpurun_out/r3p; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > "$out/pytest.log"
tail -5 "$out/pytest.log"
timeout 300 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; cut -c1-1500 "$out/bench.json"
