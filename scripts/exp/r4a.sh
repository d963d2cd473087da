#!/bin/bash
# touch-ahead of the next round's home cells: A/B against the same source without it and against HEAD
set -u
out=$PWD/gpurun_out/r4a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
L=limitador_amd/lib
cp $L/librl_engine.so $L/alt/librl_engine_touch.so
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
timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_bucketed.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
for v in touch notouch head touch; do
  cp $L/alt/librl_engine_$v.so $L/librl_engine.so
  run ${v}_20 X=1 -- --steps 20 --warmup 5
  run ${v}_200 X=1 -- --steps 200 --warmup 5
  run ${v}_uniform X=1 -- --steps 100 --warmup 5 --zipf 0
done
cp $L/alt/librl_engine_touch.so $L/librl_engine.so
