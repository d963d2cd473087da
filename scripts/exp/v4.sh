#!/bin/bash
# GPU visit: the variant tests, timed launches with their own start / stop events, the routed step.
set -u
out=$PWD/gpurun_out/v4; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --secondary 0"
timeout 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_bucketed.py -m gpu -q -x 2>&1 | tail -12 > "$out/pytest_variants.log"
timeout 200 $B > "$out/bench_a.json" 2> "$out/bench_a.err"; echo "exit $?" >> "$out/bench_a.err"
timeout 200 $B --force-sharded > "$out/bench_sharded.json" 2> "$out/bench_sharded.err"; echo "exit $?" >> "$out/bench_sharded.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/bench_under_trace.json" 2> "$out/trace.err"
cd "$OLDPWD"
find "$out" -type f -size +6M -delete
for f in "$out"/pytest_*.log; do echo "== $f"; tail -6 "$f"; done
tail -5 "$out/bench_sharded.err"
for f in "$out"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d.get("pipeline",{})
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p.get("kernel_ms_per_batch",{}).items()}, "alone", round((d["roofline"].get("avg_launch_ms_alone") or 0)*1e3,1), "frac", round(d["roofline"]["frac"],4))
except Exception as ex:
    print("ERR",ex)
PY
done
grep "k_bkt" "$out"/trace/t_kernel_stats.csv | cut -c1-160
