#!/bin/bash
# r11f — the replay's rounds read the NEXT round's home cells ahead (real loads, not a touch): parity, then the bench with the
# 80-register kernel (7 registers in scratch) and the 96-register one, read-ahead on / off.
set -u
out=$PWD/gpurun_out/r11f; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q 2>&1 | tail -8 > "$out/pytest.log"; echo "pytest exit: ${PIPESTATUS[0]}"; tail -n 3 "$out/pytest.log"
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step frac", round(d["roofline"]["frac"],3), "launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
for cfg in "1 0" "1 3" "0 0" "0 3" "1 0" "1 3"; do
  set -- $cfg
  for steps in 20 200; do
    RL_READ_AHEAD=$1 RL_APPLY2_CFG=$2 bench --steps $steps --warmup 6 > "$out/ra$1_c$2_s$steps.json" 2> "$out/ra$1_c$2_s$steps.err"
    line "$out/ra$1_c$2_s$steps.json" "read_ahead=$1 cfg=$2 steps=$steps"
  done
done
for cfg in "1 0" "1 3"; do
  set -- $cfg
  RL_READ_AHEAD=$1 RL_APPLY2_CFG=$2 bench --steps 50 --warmup 6 --depth 1 > "$out/ra$1_c$2_alone.json" 2> "$out/ra$1_c$2_alone.err"
  line "$out/ra$1_c$2_alone.json" "read_ahead=$1 cfg=$2 alone(depth 1)"
  RL_READ_AHEAD=$1 RL_APPLY2_CFG=$2 RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=60 RL_APPLY_TRACE_FILE=$out/t.bin bench --steps 100 --warmup 5 > "$out/ra$1_c$2.trace.json" 2> "$out/ra$1_c$2.trace.err"
  python scripts/apply_trace.py "$out/t.bin" > "$out/ra$1_c$2.txt" 2>&1; grep -E "hot workgroups|span|^total|^rounds" "$out/ra$1_c$2.txt"
  rm -f "$out/t.bin"
done
