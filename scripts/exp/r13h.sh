#!/bin/bash
# r13f — k_route_one with the parallel scan: parity of the route kernels, the routed step at world 1, one timeline
set -u
out=$PWD/gpurun_out/r13h; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_sharded_abi.py tests/test_gpu_sharded_procs.py tests/test_gpu_sharded.py -q -x > "$out/route.log" 2>&1; echo "sharded tests exit: $?"; tail -n 1 "$out/route.log" | cut -c1-200
export LIMITADOR_AMD_LIB=exp
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 --force-sharded "$@"; }
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/slice")
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
for cfg in "external 0" "own 0"; do
  set -- $cfg
  for steps in 20 200; do
    f="$out/$1_one$2_s$steps.json"
    RL_SHARDED_ENGINE_STREAMS=$1 RL_ROUTE_ONE=$2 bench --steps $steps --warmup 6 > "$f" 2> "${f%.json}.err"; show "$f" "engine streams=$1 route_one=$2 steps=$steps"
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_external" -o t -- python $OLDPWD/bench.py --cpu-seconds 0 --secondary 0 --force-sharded --steps 60 --warmup 6 > "$out/trace_external.json" 2> "$out/trace_external.err"
f=$(find "$out/trace_external" -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $OLDPWD/scripts/timeline.py "$f" 24 | cut -c1-110
cd "$OLDPWD"
RL_SHARDED_TRACE=1 bench --steps 12 --warmup 4 2> "$out/host_trace.err" > /dev/null; grep '^\[sh\]' "$out/host_trace.err" | tail -n 36
find "$out" -type f -size +6M -delete
