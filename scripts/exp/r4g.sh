#!/bin/bash
set -u
out=$PWD/gpurun_out/r4g; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "scenarios or per_request or registered" > $out/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert|FAILED" $out/pytest.log | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4g/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e9,2), "G/s", d["ms_per_step"])
for k,v in d["secondary"].items():
    if k.startswith("configs0") or k.startswith("host"):
        print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
PY
