#!/bin/bash
# is the replay stream's idle time between launches the write-back of the partition's dirty lines?
set -u
out=$PWD/gpurun_out/r4b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
L=limitador_amd/lib
cp $L/alt/librl_engine_exp.so $L/librl_engine.so
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
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
}
for rep in 1 2; do
run base_$rep X=1 -- --steps 200 --warmup 10
run nostore_$rep RL_PART_NOSTORE=8 -- --steps 200 --warmup 10
run wthru_$rep RL_PART_WTHRU=1 -- --steps 200 --warmup 10
done
