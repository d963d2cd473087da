#!/bin/bash
# r11l — the replay's home-cell reads as nontemporal loads (variant library, scripts/exp/patches/cell_nt_load.patch): timing only.
set -u
out=$PWD/gpurun_out/r11l; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cp limitador_amd/lib/librl_engine.so /tmp/tree.so
run() { for steps in 20 200; do timeout 60 python bench.py --cpu-seconds 0 --secondary 0 --steps $steps --warmup 6 > "$out/$1_s$steps.json" 2> "$out/$1_s$steps.err"
  python - "$out/$1_s$steps.json" "$1 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "alone", round(d["roofline"]["avg_launch_ms_alone"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done; }
run tree
cp limitador_amd/lib/variants/librl_engine_cell_nt_load.so limitador_amd/lib/librl_engine.so
run nt
cp /tmp/tree.so limitador_amd/lib/librl_engine.so
