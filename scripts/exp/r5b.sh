#!/bin/bash
# r5b: does evening the hash buckets out earlier (RL_HOT_LONG below 1024, with the small-step threshold so that the hot
# buckets fill) shorten the replay's tail?  + the phase stamps of a steady-state replay in a blocking call.
set -u
out=$PWD/gpurun_out/r5b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "idle", round((p.get("apply_stream_idle_ms_per_batch") or 0)*1e3,1))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env RL_HOT_REPORT=1 "${envs[@]}" timeout 60 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
  grep "^\[hot\]" "$out/$name.err" | tail -1
}
run trace RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace.bin RL_APPLY_TRACE_AT=14 -- --steps 20 --warmup 5
for rep in 1 2; do
run base_$rep RL_X=0 -- --steps 200 --warmup 10
run long896_$rep RL_HOT_LONG=896 -- --steps 200 --warmup 10
run long768_$rep RL_HOT_LONG=768 -- --steps 200 --warmup 10
run a_long768_$rep RL_HOT_ADAPT=1 RL_HOT_PROMOTE=64 RL_HOT_LONG=768 -- --steps 200 --warmup 10
run a_long640_$rep RL_HOT_ADAPT=1 RL_HOT_PROMOTE=64 RL_HOT_LONG=640 -- --steps 200 --warmup 10
run a96_long768_$rep RL_HOT_ADAPT=1 RL_HOT_PROMOTE=96 RL_HOT_LONG=768 RL_HOT_ADAPT_LO=448 RL_HOT_ADAPT_HI=500 -- --steps 200 --warmup 10
done
find "$out" -type f -size +8M -delete
