#!/bin/bash
# r11a — long buckets split in two (RL_SPLIT, experiment build).  Parity first (the pipeline tests with the split on), then the
# bench with and without it, and at two thresholds.
set -u
out=$PWD/gpurun_out/r11a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
RL_SPLIT=1 RL_SPLIT_MIN=300 timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q 2>&1 | tail -8 > "$out/pytest_split.log"; echo "split pytest exit: ${PIPESTATUS[0]}"; tail -n 4 "$out/pytest_split.log"
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for cfg in "0 704" "1 704" "1 640" "1 768" "1 560"; do
  set -- $cfg
  for steps in 20 200; do
    RL_SPLIT=$1 RL_SPLIT_MIN=$2 bench --steps $steps --warmup 6 > "$out/s$1_m$2_s$steps.json" 2> "$out/s$1_m$2_s$steps.err"
    python - "$out/s$1_m$2_s$steps.json" "split=$1 min=$2 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step frac", round(d["roofline"]["frac"],3), "launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "hits/launch", d["roofline"]["hits_per_launch"], "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
