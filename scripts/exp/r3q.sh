#!/bin/bash
set -u
out=$PWD/gpurun_out/r3q; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_merge.py tests/test_gpu_bucketed.py tests/test_gpu_sharded_abi.py tests/test_gpu_sharded.py tests/test_gpu_host_mirror.py -m gpu -q -x 2>&1 | tail -15
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", d["config"]["parallelism"])
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"; tail -2 "$out/$name.err"
}
run sh_ext X=1 -- --steps 100 --warmup 5 --force-sharded
run sh_own RL_SHARDED_ENGINE_STREAMS=own -- --steps 100 --warmup 5 --force-sharded
run sh_torch X=1 -- --steps 100 --warmup 5 --force-sharded --sharded-impl torch
