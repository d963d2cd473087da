#!/bin/bash
set -u
out=$PWD/gpurun_out/r4j; rm -rf "$out"; mkdir -p "$out"
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
  env "${envs[@]}" timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
}
for d in 0 6 12 18 25 35; do
  run d${d}_200 RL_PART_DELAY_US=$d -- --steps 200 --warmup 10
done
run d0_20 RL_PART_DELAY_US=0 -- --steps 20 --warmup 5
run d12_20 RL_PART_DELAY_US=12 -- --steps 20 --warmup 5
