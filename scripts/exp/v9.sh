#!/bin/bash
set -u
out=$PWD/gpurun_out/v9; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for r in 1 2; do timeout 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 > "$out/bench_$r.json" 2> "$out/bench_$r.err"; done
for f in "$out"/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "alone", round((d["roofline"].get("avg_launch_ms_alone") or 0)*1e3,1), "frac", round(d["roofline"]["frac"],4))
PY
done
