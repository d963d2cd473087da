#!/bin/bash
# GPU visit: 1024 hash buckets (k_bkt_apply leaves half of every CU's wave slots free; k_bkt_scatter's dynamic LDS
# then fits beside it) against 2048.
set -u
out=$PWD/gpurun_out/v5; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --secondary 0"
timeout 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_bucketed.py tests/test_gpu_parity.py -m gpu -q -x -k "not config3" 2>&1 | tail -5 > "$out/pytest_b11.log"
RL_BUCKET_LOG2=10 timeout 300 python -m pytest tests/test_gpu_bucketed.py -m gpu -q -x 2>&1 | tail -5 > "$out/pytest_b10.log"
timeout 200 $B > "$out/bench_b11.json" 2> "$out/bench_b11.err"
RL_BUCKET_LOG2=10 timeout 200 $B > "$out/bench_b10.json" 2> "$out/bench_b10.err"
RL_BUCKET_LOG2=10 RL_OVERLAP=0 timeout 200 $B > "$out/bench_b10_1s.json" 2> "$out/bench_b10_1s.err"
RL_BUCKET_LOG2=9 timeout 200 $B > "$out/bench_b9.json" 2> "$out/bench_b9.err"
cd /tmp
RL_BUCKET_LOG2=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/bench_under_trace.json" 2> "$out/trace.err"
cd "$OLDPWD"
find "$out" -type f -size +6M -delete
for f in "$out"/pytest_*.log; do echo "== $f"; tail -3 "$f"; done
for f in "$out"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d.get("pipeline",{})
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p.get("kernel_ms_per_batch",{}).items()}, "alone", round((d["roofline"].get("avg_launch_ms_alone") or 0)*1e3,1), "frac", round(d["roofline"]["frac"],4))
except Exception as ex:
    print("ERR",ex)
PY
done
