#!/bin/bash
set -u
mkdir -p gpurun_out
export RL_APPLY_TRACE=1 RL_APPLY_TRACE_DUMP=1
echo "== uniform 1M"; timeout 300 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --zipf 0 2>&1 | grep -E "apply trace" | tail -3
echo "== zipf 1M"; timeout 300 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 2>&1 | grep -E "apply trace" | tail -3
echo "== uniform 64k/1M keys"; timeout 300 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --zipf 0 --keys 1048576 --batch 65536 2>&1 | grep -E "apply trace" | tail -3
