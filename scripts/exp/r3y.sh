#!/bin/bash
set -u
out=$PWD/gpurun_out/r3y; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
RL_PART_COMPACT=1 timeout 400 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "idle", round((p.get("apply_stream_idle_ms_per_batch") or 0)*1e3,1), "denied", d["config"]["denied_in_last_batch"])
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
run base X=1 -- --steps 100 --warmup 5
run compact RL_PART_COMPACT=1 -- --steps 100 --warmup 5
run compact20 RL_PART_COMPACT=1 -- --steps 20 --warmup 5
run compact_hw128 RL_PART_COMPACT=1 RL_HOT_WGS=128 -- --steps 100 --warmup 5
run compact_v64 RL_PART_COMPACT=1 RL_APPLY2_CFG=2 -- --steps 100 --warmup 5
run compact_uniform RL_PART_COMPACT=1 -- --steps 100 --warmup 5 --zipf 0
run compact_cfg1 RL_PART_COMPACT=1 -- --steps 100 --warmup 5 --keys 1000000 --batch 65536 --zipf 0
run cfg1 X=1 -- --steps 100 --warmup 5 --keys 1000000 --batch 65536 --zipf 0
