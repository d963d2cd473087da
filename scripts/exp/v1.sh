#!/bin/bash
# GPU visit: pipeline depth 2/3 x k_hot_state / self_hot — parity subset, bench, SQ counters, timeline.
set -u
out=$PWD/gpurun_out/v1; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --secondary 0"
# 1. full GPU suite in the new default (depth 3, self_hot)
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > "$out/pytest_default.log"
# 2. the old configuration and the mixed ones on the hot-path tests
for cfg in "2 0"; do
  set -- $cfg
  RL_PIPE_DEPTH=$1 RL_SELF_HOT=$2 timeout 300 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py -m gpu -q -x -k "not config3" 2>&1 | tail -4 > "$out/pytest_d$1_s$2.log"
done
# 3. bench, four combinations
for cfg in "2 0" "3 0" "2 1" "3 1"; do
  set -- $cfg
  RL_PIPE_DEPTH=$1 RL_SELF_HOT=$2 timeout 200 $B > "$out/bench_d$1_s$2.json" 2> "$out/bench_d$1_s$2.err"
done
cd /tmp
# 4. timeline of the default
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/bench_under_trace.json" 2> "$out/trace.err"
# 5. SQ counters of the engine's kernels (own runs)
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo $pmc | tr ' ' '+')
  RL_OVERLAP=0 timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/sq_$name" -o p -- $B > /dev/null 2> "$out/sq_$name.err"
done
cd "$OLDPWD"
find "$out" -type f -size +6M -delete
for f in "$out"/pytest_*.log; do echo "== $f"; tail -3 "$f"; done
for f in "$out"/bench_d*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "host_submit", round(d["pipeline"]["host_submit_us_per_batch"],1))
except Exception as ex:
    print("ERR",ex)
PY
done
