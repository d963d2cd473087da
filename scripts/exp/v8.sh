#!/bin/bash
# GPU visit: how many keys should get a hot bucket (floor of the promotion threshold; the long-bucket rule).
set -u
out=$PWD/gpurun_out/v8; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 200 --warmup 10 --cpu-seconds 0 --secondary 0"
for f in 48 80 112 128 160; do RL_HOT_PROMOTE=$f timeout 200 $B > "$out/bench_f$f.json" 2> "$out/bench_f$f.err"; done
RL_HOT_LONG=1024 timeout 200 $B > "$out/bench_long1024.json" 2> "$out/bench_long1024.err"
RL_HOT_LONG=1024 RL_HOT_PROMOTE=80 timeout 200 $B > "$out/bench_long1024_f80.json" 2> "$out/bench_long1024_f80.err"
for f in "$out"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d.get("pipeline",{})
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p.get("kernel_ms_per_batch",{}).items()}, "alone", round((d["roofline"].get("avg_launch_ms_alone") or 0)*1e3,1))
except Exception as ex:
    print("ERR",ex)
PY
done
