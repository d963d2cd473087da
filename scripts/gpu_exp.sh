#!/bin/bash
set -u
mkdir -p gpurun_out
export RL_APPLY_TRACE=1 RL_APPLY_TRACE_DUMP=1
for cm in 1 4; do
echo "== zipf 1M cap-mult $cm"; timeout 300 python bench.py --steps 4 --warmup 3 --cpu-seconds 0 --cap-mult $cm 2>&1 | grep -E "apply trace|metric" | tail -3 | cut -c1-400
done
