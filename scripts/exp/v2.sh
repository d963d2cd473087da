#!/bin/bash
# GPU visit: k_bkt_apply with 19.5 KB of LDS (eight workgroups per CU) against the 21 KB form; where its time goes
# (RL_DEBUG_APPLY2); table at load 0.15; the matcher's round trip under the fill pass.
set -u
out=$PWD/gpurun_out/v2; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --secondary 0"
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > "$out/pytest_default.log"
RL_APPLY2_CFG=1 timeout 300 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py -m gpu -q -x -k "not config3" 2>&1 | tail -4 > "$out/pytest_wide.log"
timeout 200 $B > "$out/bench_narrow.json" 2> "$out/bench_narrow.err"
RL_APPLY2_CFG=1 timeout 200 $B > "$out/bench_wide.json" 2> "$out/bench_wide.err"
timeout 200 $B --cap-mult 2 > "$out/bench_cap2.json" 2> "$out/bench_cap2.err"
for d in 2 4 8 14; do RL_DEBUG_APPLY2=$d timeout 200 $B > "$out/bench_dbg$d.json" 2> "$out/bench_dbg$d.err"; done
RL_OVERLAP=0 timeout 200 $B > "$out/bench_1stream.json" 2> "$out/bench_1stream.err"
timeout 200 python scripts/bench_match.py --steps 10 > "$out/match.json" 2> "$out/match.err"
for f in "$out"/pytest_*.log; do echo "== $f"; tail -3 "$f"; done
for f in "$out"/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "alone", round(d["roofline"]["avg_launch_ms_alone"]*1e3,1), "host_submit", round(d["pipeline"]["host_submit_us_per_batch"],1))
except Exception as ex:
    print("ERR",ex)
PY
done
cat "$out/match.json"
