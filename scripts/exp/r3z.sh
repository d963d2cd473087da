#!/bin/bash
set -u
out=$PWD/gpurun_out/r3z; rm -rf "$out"; mkdir -p "$out"
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
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
}
for rep in 1 2; do
run c0_20_$rep X=1 -- --steps 20 --warmup 5
run c2_20_$rep RL_APPLY2_CFG=2 -- --steps 20 --warmup 5
run c0_1000_$rep X=1 -- --steps 1000 --warmup 5
run c2_1000_$rep RL_APPLY2_CFG=2 -- --steps 1000 --warmup 5
done
run c2_hw128 RL_APPLY2_CFG=2 RL_HOT_WGS=128 -- --steps 100 --warmup 5
run c3 RL_APPLY2_CFG=3 -- --steps 100 --warmup 5
run c0_uniform X=1 -- --steps 100 --warmup 5 --zipf 0
run c2_uniform RL_APPLY2_CFG=2 -- --steps 100 --warmup 5 --zipf 0
