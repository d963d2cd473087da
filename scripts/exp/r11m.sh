#!/bin/bash
# r11m — the replay's workgroups out of step (scripts/exp/patches/stagger_24.patch / _48: a quarter of the bucket workgroups each
# start their rounds 0 / 1 / 2 / 3 x 0.65 us (1.3 us) late, so that the rounds' cell reads do not all arrive at once): timing only.
set -u
out=$PWD/gpurun_out/r11m; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cp limitador_amd/lib/librl_engine.so /tmp/tree.so
run() { for steps in 200; do timeout 40 python bench.py --cpu-seconds 0 --secondary 0 --steps $steps --warmup 6 > "$out/$1_s$steps.json" 2> "$out/$1_s$steps.err"
  python - "$out/$1_s$steps.json" "$1 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "alone", round(d["roofline"]["avg_launch_ms_alone"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done; }
run tree
for v in 24 48; do cp limitador_amd/lib/variants/librl_engine_stagger_$v.so limitador_amd/lib/librl_engine.so; run stagger_$v; done
cp /tmp/tree.so limitador_amd/lib/librl_engine.so
