#!/bin/bash
set -u
out=$PWD/gpurun_out/r3s; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > "$out/bench.json" 2> "$out/bench.err"
python - "$out/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step")
for k,v in d["secondary"].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
PY
