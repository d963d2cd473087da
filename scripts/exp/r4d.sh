#!/bin/bash
set -u
out=$PWD/gpurun_out/r4d; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py tests/test_gpu_rls_e2e.py tests/test_gpu_sharded_multi.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4d/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e9,2), "G/s", d["ms_per_step"])
for k,v in d["secondary"].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
PY
