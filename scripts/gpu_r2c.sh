#!/bin/bash
# round-2 visit C: where does k_bkt_apply2's time go?  (timing experiments: results are wrong by design)
set -u
mkdir -p gpurun_out/r2c
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
for dbg in 0 1 2 4 8 3 7 15; do
  RL_DEBUG_APPLY2=$dbg timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2c/b_$dbg.json 2> gpurun_out/r2c/b_$dbg.err; short gpurun_out/r2c/b_$dbg.json "dbg=$dbg"
done
for dbg in 0 1 2 4 8 15; do
  RL_DEBUG_APPLY2=$dbg timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 > gpurun_out/r2c/u_$dbg.json 2> gpurun_out/r2c/u_$dbg.err; short gpurun_out/r2c/u_$dbg.json "uniform dbg=$dbg"
done
