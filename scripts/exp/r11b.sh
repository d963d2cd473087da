#!/bin/bash
# r11b — where the replay's span goes with long buckets split (RL_SPLIT 0 / 1 = hot workgroups give theirs up / 2 = extra
# workgroups on top): raw stamps of one steady-state batch each (scripts/apply_trace.py), then 200-step benches.
set -u
out=$PWD/gpurun_out/r11b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for cfg in "0 704" "1 704" "2 704" "2 560" "1 900"; do
  set -- $cfg
  tag=s$1_m$2
  RL_SPLIT=$1 RL_SPLIT_MIN=$2 RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=60 RL_APPLY_TRACE_FILE=$out/$tag.bin bench --steps 100 --warmup 5 > "$out/$tag.trace.json" 2> "$out/$tag.trace.err"
  echo "== split=$1 min=$2"; grep "^\[apply\]" "$out/$tag.trace.err" | tail -1 | cut -c1-260
  python scripts/apply_trace.py "$out/$tag.bin" > "$out/$tag.txt" 2>&1; grep -E "hot workgroups|span|^total|^end|^hits|^rounds" "$out/$tag.txt"
  for steps in 20 200; do
    RL_SPLIT=$1 RL_SPLIT_MIN=$2 bench --steps $steps --warmup 6 > "$out/${tag}_s$steps.json" 2> "$out/${tag}_s$steps.err"
    python - "$out/${tag}_s$steps.json" "split=$1 min=$2 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step frac", round(d["roofline"]["frac"],3), "launch", round(d["roofline"]["avg_launch_ms"]*1e3,1))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
  rm -f "$out/$tag.bin"
done
