#!/bin/bash
# r11j — the partition's records written through (RL_PART_WT=1: `sc1` 16-byte stores, 2: nontemporal; experiment build): the L2
# write-back at the end of a kernel sits in front of the next launch of the stream (scripts/microbench/kernel_gap.hip).
set -u
out=$PWD/gpurun_out/r11j; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
bench() { timeout 100 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for wt in 0 1 2 1 0; do
  for steps in 20 200; do
    RL_PART_WT=$wt bench --steps $steps --warmup 6 > "$out/wt${wt}_s$steps.json" 2> "$out/wt${wt}_s$steps.err"
    python - "$out/wt${wt}_s$steps.json" "wt=$wt steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "idle", round(p["apply_stream_idle_ms_per_batch"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
RL_PART_WT=1 timeout 200 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -x -q 2>&1 | tail -3
