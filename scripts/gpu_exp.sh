#!/bin/bash
export RL_APPLY_TRACE=1 RL_APPLY_TRACE_DUMP=1
timeout 300 python bench.py --steps 3 --warmup 3 --cpu-seconds 0 --depth 1 2>&1 | grep -E "apply trace" | tail -1 | grep -o "scatter: .*rowwrite=[0-9.]*"
