#!/bin/bash
# r13d — the routed step at world 1 (bench.py --force-sharded: RCCL communicator of one rank, the whole router path) now that
# the engine holds replays back (RL_DEFER2): the communicator's one apply stream against the engine's own two streams, and the
# kernel timeline of both (rocprofv3 --kernel-trace) for scripts/timeline.py.  First: the suites touched since r13c.
set -u
out=$PWD/gpurun_out/r13e; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_sharded.py tests/test_gpu_sharded_abi.py tests/test_gpu_sharded_procs.py -q -x > "$out/touched.log" 2>&1; rc=$?
echo "touched suites exit: $rc"; tail -n 2 "$out/touched.log" | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" "$out/touched.log" | head -30; fi
export LIMITADOR_AMD_LIB=exp
bench() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 --force-sharded "$@"; }
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/slice", d["config"]["parallelism"][:60])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
for mode in external own; do
  for steps in 20 200; do
    f="$out/${mode}_s$steps.json"
    RL_SHARDED_ENGINE_STREAMS=$mode bench --steps $steps --warmup 6 > "$f" 2> "${f%.json}.err"; show "$f" "engine streams=$mode steps=$steps"
  done
done
cd /tmp
for mode in external own; do
  RL_SHARDED_ENGINE_STREAMS=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$mode" -o t -- python $OLDPWD/bench.py --cpu-seconds 0 --secondary 0 --force-sharded --steps 60 --warmup 6 > "$out/trace_$mode.json" 2> "$out/trace_$mode.err"
  f=$(find "$out/trace_$mode" -name '*kernel_trace.csv' | head -1)
  echo "== timeline, engine streams=$mode ($f)"
  [ -n "$f" ] && python $OLDPWD/scripts/timeline.py "$f" 48 | cut -c1-110
done
cd "$OLDPWD"
RL_SHARDED_TRACE=1 RL_SHARDED_ENGINE_STREAMS=own bench --steps 12 --warmup 4 2> "$out/host_trace_own.err" > /dev/null; grep '^\[sh\]' "$out/host_trace_own.err" | tail -n 60 > "$out/host_trace_own.txt"; wc -l "$out/host_trace_own.txt"
find "$out" -type f -size +6M -delete
