#!/bin/bash
# r9a — first visit of the session: the GPU suite at the tree (in-place compaction, expiry legs of the bench), the bench
# line with its secondary block, rocprofv3 kernel stats of the streaming maintenance kernels (k_scan<3>, k_compact_*).
set -u
out=$PWD/gpurun_out/r9a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$out/pytest_gpu.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench exit: $?" >> "$out/bench.err"
timeout 300 python scripts/bench_sweep.py > "$out/sweep.json" 2> "$out/sweep.err"
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/sweep_trace" -o t -- python $OLDPWD/scripts/bench_sweep.py --reps 5 > "$out/sweep_under_trace.json" 2> "$out/sweep_trace.err" )
find "$out" -type f -size +4M -delete
tail -n 5 "$out/pytest_gpu.log"; tail -n 3 "$out/bench.err"; cat "$out/sweep.json"
python - "$out/bench.json" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print(d["value"]/1e9, d["ms_per_step"], d["roofline"]["frac"])
s=d["secondary"]
for k in ("headline_1000_steps","headline_with_expiry","uniform_10M_keys_1M_hits","configs4_shape_match_and_check_1M_requests","sweep_and_compact_10M_keys"):
    print(k, json.dumps(s.get(k))[:900])
PY
