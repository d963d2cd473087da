#!/bin/bash
# r11k — the hot threshold and the hot workgroups again, now that the table is at load 0.15 (experiment build; 200-step benches).
set -u
out=$PWD/gpurun_out/r11k; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for cfg in "160 256" "136 256" "120 256" "200 256" "136 320" "120 320" "160 320" "160 192" "120 384" "160 256"; do
  set -- $cfg
  RL_HOT_PROMOTE=$1 RL_HOT_WGS=$2 timeout 60 python bench.py --cpu-seconds 0 --secondary 0 --steps 200 --warmup 8 > "$out/p$1_w$2.json" 2> "$out/p$1_w$2.err"
  python - "$out/p$1_w$2.json" "promote=$1 hot_wgs=$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
