#!/bin/bash
set -u
out=$PWD/gpurun_out/r3x; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
python - "$out/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", "frac", round(d["roofline"]["frac"],4), d["pipeline"]["kernel_ms_per_batch_in_pipeline"], d["pipeline"]["apply_stream_idle_ms_per_batch"])
for k,v in d["secondary"].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["threads_tried"])
PY
