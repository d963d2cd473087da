#!/bin/bash
set -u
mkdir -p gpurun_out/r2e
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
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bucketed.py -m gpu -x -q 2>&1 | tail -4
for dbg in 0 16 32 1 0 16 32; do
  RL_DEBUG_APPLY2=$dbg timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2e/b_$dbg.json 2> gpurun_out/r2e/b_$dbg.err; short gpurun_out/r2e/b_$dbg.json "dbg=$dbg"
done
