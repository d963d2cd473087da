#!/bin/bash
# r5a: what the hot set looks like in steady state (keys that qualify, threshold, bucket lengths of one steady-state
# replay), and whether a threshold that moves in small steps towards a nearly full set (RL_HOT_ADAPT=1) shortens the step.
set -u
out=$PWD/gpurun_out/r5a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "idle", round((p.get("apply_stream_idle_ms_per_batch") or 0)*1e3,1))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env RL_HOT_REPORT=1 "${envs[@]}" timeout 100 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
  grep "^\[hot\]" "$out/$name.err" | tail -3
}
run trace RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace.bin RL_APPLY_TRACE_AT=30 -- --steps 20 --warmup 5
grep "^\[apply\]" "$out/trace.err" | tail -2
run base_200 RL_X=0 -- --steps 200 --warmup 10
run adapt64_200 RL_HOT_ADAPT=1 RL_HOT_PROMOTE=64 -- --steps 200 --warmup 10
run adapt64_20 RL_HOT_ADAPT=1 RL_HOT_PROMOTE=64 -- --steps 20 --warmup 5
run adapt64w_200 RL_HOT_ADAPT=1 RL_HOT_PROMOTE=64 RL_HOT_ADAPT_LO=352 RL_HOT_ADAPT_HI=448 -- --steps 200 --warmup 10
run p112_200 RL_HOT_PROMOTE=112 -- --steps 200 --warmup 10
run base_20 RL_X=0 -- --steps 20 --warmup 5
find "$out" -type f -size +8M -delete
