#!/bin/bash
# r9j — the routed step with a rank's own segment kept off the communicator: sharded tests, the bench line, the engine's two streams under the router once more.
set -u
out=$PWD/gpurun_out/r9j; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sharded_abi.py tests/test_gpu_sharded.py tests/test_gpu_sharded_multi.py -x -q 2>&1 | tail -3
for mode in default own; do
  if [ $mode = own ]; then export LIMITADOR_AMD_LIB=exp RL_SHARDED_ENGINE_STREAMS=own; fi
  timeout 300 python bench.py --force-sharded --steps 100 --warmup 10 --cpu-seconds 0 --secondary 0 > "$out/bench_$mode.json" 2> "$out/bench_$mode.err"
  python - "$out/bench_$mode.json" $mode <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print(sys.argv[2], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/slice")
PY
done
unset LIMITADOR_AMD_LIB RL_SHARDED_ENGINE_STREAMS
timeout 120 python scripts/bench_sharded_requests.py 2>/dev/null | cut -c1-400 | tail -2
