#!/bin/bash
# GPU visit: verdicts prefilled by k_bkt_hist, k_bkt_apply stores only the denials.
set -u
out=$PWD/gpurun_out/v3; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --secondary 0"
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > "$out/pytest_default.log"
timeout 200 $B > "$out/bench_a.json" 2> "$out/bench_a.err"
timeout 200 $B --steps 1000 > "$out/bench_1000.json" 2> "$out/bench_1000.err"
timeout 200 $B --zipf 0 > "$out/bench_uniform.json" 2> "$out/bench_uniform.err"
timeout 200 $B --force-sharded > "$out/bench_sharded.json" 2> "$out/bench_sharded.err"
for f in "$out"/pytest_*.log; do echo "== $f"; tail -3 "$f"; done
for f in "$out"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d.get("pipeline",{})
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p.get("kernel_ms_per_batch",{}).items()}, "alone", round(d["roofline"].get("avg_launch_ms_alone",0)*1e3,1))
except Exception as ex:
    print("ERR",ex)
PY
done
