#!/bin/bash
# round-2 visit D: sharded ticket; store-width / 32-byte-cell microbenchmarks
set -u
mkdir -p gpurun_out/r2d
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
for c in 0 2 3; do
  RL_APPLY2_CFG=$c timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2d/b_c$c.json 2> gpurun_out/r2d/b_c$c.err; short gpurun_out/r2d/b_c$c.json "cfg=$c"
done
RL_APPLY=1 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2d/b_v1.json 2> gpurun_out/r2d/b_v1.err; short gpurun_out/r2d/b_v1.json "v1"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 > gpurun_out/r2d/u.json 2> gpurun_out/r2d/u.err; short gpurun_out/r2d/u.json "uniform"
timeout 300 scripts/microbench/bin/random_access > gpurun_out/r2d/microbench.txt 2>&1; cat gpurun_out/r2d/microbench.txt
