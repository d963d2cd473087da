#!/bin/bash
set -u
mkdir -p gpurun_out/r2i
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
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 5 --cpu-seconds 0 > gpurun_out/r2i/$name.json 2> gpurun_out/r2i/$name.err; short gpurun_out/r2i/$name.json "$name"
}
run base X=1
run bk10_c0 RL_BUCKET_LOG2=10
run bk10_c3 RL_BUCKET_LOG2=10 RL_APPLY2_CFG=3
run bk10_c2 RL_BUCKET_LOG2=10 RL_APPLY2_CFG=2
run bk11_c3 RL_APPLY2_CFG=3
run nooverlap RL_OVERLAP=0
run bk10_c3_noov RL_BUCKET_LOG2=10 RL_APPLY2_CFG=3 RL_OVERLAP=0
