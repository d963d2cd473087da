#!/bin/bash
# GPU visit: 1024 buckets by default, cross-stream events as the launches' own stop events.
set -u
out=$PWD/gpurun_out/v7; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --secondary 0"
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > "$out/pytest_default.log"
timeout 200 $B > "$out/bench_a.json" 2> "$out/bench_a.err"
RL_APPLY2_CFG=1 timeout 200 $B > "$out/bench_wide.json" 2> "$out/bench_wide.err"
timeout 200 $B --steps 1000 > "$out/bench_1000.json" 2> "$out/bench_1000.err"
timeout 200 $B --zipf 0 > "$out/bench_uniform.json" 2> "$out/bench_uniform.err"
RL_BUCKET_LOG2=11 timeout 200 $B > "$out/bench_b11.json" 2> "$out/bench_b11.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/bench_under_trace.json" 2> "$out/trace.err"
cd "$OLDPWD"
find "$out" -type f -size +6M -delete
for f in "$out"/pytest_*.log; do echo "== $f"; tail -3 "$f"; done
for f in "$out"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d.get("pipeline",{})
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p.get("kernel_ms_per_batch",{}).items()}, "alone", round((d["roofline"].get("avg_launch_ms_alone") or 0)*1e3,1), "frac", round(d["roofline"]["frac"],4), "host", p.get("host_submit_us_per_batch"))
except Exception as ex:
    print("ERR",ex)
PY
done
