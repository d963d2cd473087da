#!/bin/bash
# routed step behind the C ABI: tests + world-1 bench (torch driver vs C entry, engine own streams vs one stream)
mkdir -p gpurun_out/r2m
timeout 600 python -m pytest tests/test_gpu_sharded_abi.py tests/test_gpu_sharded.py -q -x 2>&1 | tail -15 > gpurun_out/r2m/tests.txt
cat gpurun_out/r2m/tests.txt
for impl in torch abi; do
  timeout 300 python bench.py --force-sharded --sharded-impl $impl --steps 200 --warmup 10 --cpu-seconds 0 --secondary 0 > gpurun_out/r2m/bench_$impl.json 2> gpurun_out/r2m/bench_$impl.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2m/bench_$impl.json").read().strip().splitlines()[-1])
    print("$impl", "value %.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"])
except Exception as ex:
    print("$impl failed", ex); print(open("gpurun_out/r2m/bench_$impl.err").read()[-1500:])
PY
done
RL_SHARDED_ENGINE_STREAMS=external timeout 300 python bench.py --force-sharded --sharded-impl abi --steps 200 --warmup 10 --cpu-seconds 0 --secondary 0 > gpurun_out/r2m/bench_abi_ext.json 2> gpurun_out/r2m/bench_abi_ext.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2m/bench_abi_ext.json").read().strip().splitlines()[-1])
    print("abi external-stream", "value %.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"])
except Exception as ex:
    print("abi ext failed", ex); print(open("gpurun_out/r2m/bench_abi_ext.err").read()[-1500:])
PY
