#!/bin/bash
set -u
out=$PWD/gpurun_out/r4o; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 20 --warmup 5 > $out/b$rep.json 2>/dev/null
  python - $out/b$rep.json <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,2),"us/step")
PY
done
timeout 600 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -1
