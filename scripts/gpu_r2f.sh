#!/bin/bash
set -u
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
short() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value=%.4g ms/step=%.4f"%(d["value"],d["ms_per_step"]), {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12
for dbg in 0 16 32 1; do
  RL_DEBUG_APPLY2=$dbg timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2f/b_$dbg.json 2> gpurun_out/r2f/b_$dbg.err; short gpurun_out/r2f/b_$dbg.json "dbg=$dbg"
done
for c in 2 3; do
  RL_APPLY2_CFG=$c timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2f/b_c$c.json 2> gpurun_out/r2f/b_c$c.err; short gpurun_out/r2f/b_c$c.json "cfg=$c"
done
RL_APPLY=1 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2f/b_v1.json 2> gpurun_out/r2f/b_v1.err; short gpurun_out/r2f/b_v1.json "v1"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 > gpurun_out/r2f/u.json 2> gpurun_out/r2f/u.err; short gpurun_out/r2f/u.json "uniform"
