#!/bin/bash
# r10a — pair mode (RL_PAIR=1, experiment build): two batches per replay launch.  Parity first (the pipeline tests with the
# switch on), then the bench at depth 3 / 4 with and without it.
set -u
out=$PWD/gpurun_out/r10a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
RL_PAIR=1 timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py -x -q -k "not one_million" 2>&1 | tail -8 > "$out/pytest_pair.log"; echo "pair pytest exit: ${PIPESTATUS[0]}"; tail -n 4 "$out/pytest_pair.log"
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for cfg in "0 3" "1 3" "1 4"; do
  set -- $cfg
  for steps in 20 200; do
    RL_PAIR=$1 bench --depth $2 --steps $steps --warmup 6 > "$out/p$1_d$2_s$steps.json" 2> "$out/p$1_d$2_s$steps.err"
    python - "$out/p$1_d$2_s$steps.json" "pair=$1 depth=$2 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step frac", round(d["roofline"]["frac"],3), "launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "hits/launch", d["roofline"]["hits_per_launch"], "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
