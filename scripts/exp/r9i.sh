#!/bin/bash
# r9i — the routed step on one GPU (RCCL world 1): the bench line and the kernel timeline of its streams.
set -u
out=$PWD/gpurun_out/r9i; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python bench.py --force-sharded --steps 100 --warmup 10 --cpu-seconds 0 --secondary 0 > "$out/bench.json" 2> "$out/bench.err"
python - "$out/bench.json" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print(d["value"]/1e9, d["ms_per_step"]*1e3, d["config"].get("parallelism"))
PY
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o t -- python $OLDPWD/bench.py --force-sharded --steps 30 --warmup 10 --cpu-seconds 0 --secondary 0 > "$out/bench_prof.json" 2> "$out/trace.err" )
find "$out" -type f -size +6M -delete
python $PWD/scripts/timeline.py "$out/trace/t_kernel_trace.csv" 36 | cut -c1-120
