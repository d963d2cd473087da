#!/bin/bash
# r11c — the replay's rounds touch the next round's home cells (RL_READ_AHEAD, experiment build): parity, bench with / without,
# kernels alone (--depth 1), stamps of a steady-state batch.
set -u
out=$PWD/gpurun_out/r11c; rm -rf "$out"; mkdir -p "$out"
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
for ra in 0 1 0 1; do
  for steps in 20 200; do
    RL_READ_AHEAD=$ra bench --steps $steps --warmup 6 > "$out/ra${ra}_s$steps.json" 2> "$out/ra${ra}_s$steps.err"
    line "$out/ra${ra}_s$steps.json" "read_ahead=$ra steps=$steps"
  done
done
for ra in 0 1; do
  RL_READ_AHEAD=$ra bench --steps 50 --warmup 6 --depth 1 > "$out/ra${ra}_alone.json" 2> "$out/ra${ra}_alone.err"
  line "$out/ra${ra}_alone.json" "read_ahead=$ra alone(depth 1)"
  RL_READ_AHEAD=$ra RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=60 RL_APPLY_TRACE_FILE=$out/ra$ra.bin bench --steps 100 --warmup 5 > "$out/ra$ra.trace.json" 2> "$out/ra$ra.trace.err"
  python scripts/apply_trace.py "$out/ra$ra.bin" > "$out/ra$ra.txt" 2>&1; grep -E "hot workgroups|span|^total|^rounds" "$out/ra$ra.txt"
  rm -f "$out/ra$ra.bin"
done
