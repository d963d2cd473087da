#!/bin/bash
# round-2 visit B: parity of k_bkt_apply2, then A/B timings of its instantiations against k_bkt_apply
set -u
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bucketed.py tests/test_gpu_fuzz.py tests/test_golden_traces.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2b/pytest.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/r2b/pytest.log
tail -8 gpurun_out/r2b/pytest.log
short() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "value=%.4g ms/step=%.4f"%(d["value"],d["ms_per_step"]), {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "denied",d["config"]["denied_in_last_batch"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
RL_APPLY=1 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2b/bench_v1.json 2> gpurun_out/r2b/bench_v1.err; short gpurun_out/r2b/bench_v1.json
for c in 0 2 3; do
  RL_APPLY2_CFG=$c timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r2b/bench_c$c.json 2> gpurun_out/r2b/bench_c$c.err; short gpurun_out/r2b/bench_c$c.json
done
RL_APPLY2_CFG=0 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 > gpurun_out/r2b/bench_uni.json 2> gpurun_out/r2b/bench_uni.err; short gpurun_out/r2b/bench_uni.json
RL_APPLY2_CFG=0 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --zipf 0 --keys 1048576 --batch 65536 > gpurun_out/r2b/bench_cfg1.json 2> gpurun_out/r2b/bench_cfg1.err; short gpurun_out/r2b/bench_cfg1.json
