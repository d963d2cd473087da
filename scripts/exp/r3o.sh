#!/bin/bash
set -u
out=$PWD/gpurun_out/r3o; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "idle", round((p.get("apply_stream_idle_ms_per_batch") or 0)*1e3,1), "host", round(p.get("host_submit_us_per_batch") or 0,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"; grep "^\[engine\]" "$out/$name.err" | tail -1
}
run base RL_APPLY_EVENTS=1 -- --steps 100 --warmup 5
run nt RL_APPLY_EVENTS=1 RL_PART_NT=1 -- --steps 100 --warmup 5
run ahi RL_APPLY_EVENTS=1 RL_ASTREAM_HI=1 RL_PSTREAM_PRIO=0 -- --steps 100 --warmup 5
run ahi_nt RL_APPLY_EVENTS=1 RL_ASTREAM_HI=1 RL_PSTREAM_PRIO=0 RL_PART_NT=1 -- --steps 100 --warmup 5
run noev_nt RL_PART_NT=1 -- --steps 100 --warmup 5 --timing-mode 0
