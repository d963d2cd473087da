#!/bin/bash
set -u
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
short() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value=%.4g ms/step=%.4f"%(d["value"],d["ms_per_step"]), {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "roof", round(d["roofline"]["avg_launch_ms"]*1e3,1))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
for d in 3 2 1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --depth $d > gpurun_out/r2g/b_d$d.json 2> gpurun_out/r2g/b_d$d.err; short gpurun_out/r2g/b_d$d.json "overlap depth=$d"
done
RL_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --depth 2 > gpurun_out/r2g/b_no.json 2> gpurun_out/r2g/b_no.err; short gpurun_out/r2g/b_no.json "no overlap depth=2"
timeout 300 python bench.py --steps 200 --warmup 5 --cpu-seconds 0 --depth 3 > gpurun_out/r2g/b_long.json 2> gpurun_out/r2g/b_long.err; short gpurun_out/r2g/b_long.json "overlap depth=3 200 steps"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 > gpurun_out/r2g/u.json 2> gpurun_out/r2g/u.err; short gpurun_out/r2g/u.json "uniform"
tail -3 gpurun_out/r2g/b_d3.err
