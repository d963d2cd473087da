#!/bin/bash
set -u
out=$PWD/gpurun_out/r3e; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "idle", round((p.get("apply_stream_idle_ms_per_batch") or 0)*1e3,1), "slack", round((p.get("partition_done_before_apply_ms") or 0)*1e3,1), "host", round(p.get("host_submit_us_per_batch") or 0,1))
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
run base RL_APPLY_TRACE=0 -- --steps 100 --warmup 5
run basew RL_APPLY_TRACE=1 -- --steps 100 --warmup 5
run hw64 RL_HOT_WGS=64 -- --steps 100 --warmup 5
run hw64v80 RL_HOT_WGS=64 RL_APPLY2_CFG=2 -- --steps 100 --warmup 5
run nodefer RL_DEFER_APPLY=0 -- --steps 100 --warmup 5
run noext RL_EXT_EVENTS=0 -- --steps 100 --warmup 5
run prio0 RL_PSTREAM_PRIO=0 -- --steps 100 --warmup 5
run depth2 X=1 -- --steps 100 --warmup 5 --depth 2
