#!/bin/bash
# One GPU-box visit while iterating on the hot path: parity first, then A/B timings.
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_iter.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
short() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value=%.3g ms/step=%.4f"%(d["value"],d["ms_per_step"]), {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "denied",d["config"]["denied_in_last_batch"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for b in 11; do
  RL_BUCKET_LOG2=$b timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.err
  short gpurun_out/bench_b$b.json; tail -2 gpurun_out/bench_b$b.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 > gpurun_out/bench_uniform.json 2> gpurun_out/bench_uniform.err
short gpurun_out/bench_uniform.json
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 --keys 1048576 --batch 65536 > gpurun_out/bench_cfg1.json 2> gpurun_out/bench_cfg1.err
short gpurun_out/bench_cfg1.json
RL_APPLY_TRACE=1 RL_APPLY_TRACE_DUMP=1 timeout 300 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 2>&1 | grep -E "apply trace" | tail -2
for m in 1 4; do
RL_APPLY_WG_PER_CU=$m timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/bench_wg$m.json 2> gpurun_out/bench_wg$m.err
short gpurun_out/bench_wg$m.json
done
