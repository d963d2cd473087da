#!/bin/bash
# r11g — the table's load factor against the replay's probe chains (phase B of a round ends with the round's longest chain of
# dependent probes): --cap-mult 1 / 2 / 4 = 2^25 / 2^26 / 2^27 cells for 10 M keys (load 0.30 / 0.15 / 0.075).
set -u
out=$PWD/gpurun_out/r11g; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for cm in 1 2 4 1 2; do
  for steps in 20 200; do
    timeout 200 python bench.py --cpu-seconds 0 --secondary 0 --cap-mult $cm --steps $steps --warmup 6 > "$out/cm${cm}_s$steps.json" 2> "$out/cm${cm}_s$steps.err"
    python - "$out/cm${cm}_s$steps.json" "cap-mult=$cm steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step frac", round(d["roofline"]["frac"],3), "launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
