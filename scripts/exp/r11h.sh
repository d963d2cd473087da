#!/bin/bash
# r11h — displaced keys probe PROBE_W tags per trip (phase B of a replay round): parity, then the bench at load 0.30 / 0.15 with
# the 80-register kernel (a few registers in scratch) and the 96-register one.
set -u
out=$PWD/gpurun_out/r11h; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8 > "$out/pytest.log"; echo "pytest exit: ${PIPESTATUS[0]}"; tail -n 3 "$out/pytest.log"
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step frac", round(d["roofline"]["frac"],3), "launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
for cfg in "0 1" "3 1" "0 2" "3 2" "0 1"; do
  set -- $cfg
  for steps in 20 200; do
    RL_APPLY2_CFG=$1 bench --cap-mult $2 --steps $steps --warmup 6 > "$out/c$1_m$2_s$steps.json" 2> "$out/c$1_m$2_s$steps.err"
    line "$out/c$1_m$2_s$steps.json" "cfg=$1 cap-mult=$2 steps=$steps"
  done
done
RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=60 RL_APPLY_TRACE_FILE=$out/t.bin bench --steps 100 --warmup 5 > "$out/trace.json" 2> "$out/trace.err"
python scripts/apply_trace.py "$out/t.bin" > "$out/trace.txt" 2>&1; grep -E "hot workgroups|span|^total|^rounds" "$out/trace.txt"
rm -f "$out/t.bin"
