#!/bin/bash
# round-2 visit A: phase stamps of the three partition kernels and of k_bkt_apply (state at round start)
set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
RL_APPLY_TRACE=1 RL_APPLY_TRACE_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 3 --cpu-seconds 0 > gpurun_out/r2a/trace_bench.json 2> gpurun_out/r2a/trace.err
grep -E "apply trace" gpurun_out/r2a/trace.err | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2a/bench.json").read().strip().splitlines()[-1])
print("value=%.4g ms/step=%.4f"%(d["value"],d["ms_per_step"]), d["pipeline"]["kernel_ms_per_batch"])
PY
