#!/bin/bash
# r13a — first visit of round 5.
#   1. parity on exactly what the driver times: release library, 2^26-cell table, three in flight, timed launches
#   2. RL_DEFER2 / RL_TIMING_LAZY (prepared in round 4, never run): parity with the switch on, then the bench with / without
#   3. the SIGABRT of profiles/r04f_pytest_gpu.txt: the whole GPU suite twice with the full log kept, then the test it died in,
#      in a loop, with every device buffer's address range logged (RL_LOG_ALLOCS=1: a "Memory access fault by GPU" names an address)
set -u
out=$PWD/gpurun_out/r13a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
echo "== 1. bench configuration, release build"
LIMITADOR_AMD_LIB=release timeout 900 python -m pytest tests/test_gpu_bench_config.py -x -q > "$out/bench_config.log" 2>&1; echo "exit: $?"; tail -n 3 "$out/bench_config.log"
echo "== 2. DEFER2 parity"
export LIMITADOR_AMD_LIB=exp
RL_TEST_DEFER2=1 timeout 600 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -x -q > "$out/pytest_defer2.log" 2>&1; echo "pytest (opt-in runs) exit: $?"; tail -n 3 "$out/pytest_defer2.log"
RL_DEFER2=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "config3 or config2" > "$out/pytest_full_size.log" 2>&1; echo "pytest (full size, RL_DEFER2=1) exit: $?"; tail -n 2 "$out/pytest_full_size.log"
bench() { timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "replay stream idle", round(p["apply_stream_idle_ms_per_batch"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  for steps in 20 200; do
    f="$out/d$1_l$2_s${steps}_r$rep.json"
    RL_DEFER2=$1 RL_TIMING_LAZY=$2 bench --steps $steps --warmup 6 > "$f" 2> "${f%.json}.err"
    show "$f" "defer2=$1 timing_lazy=$2 steps=$steps rep=$rep"
  done
done
done
for d2 in 0 1; do RL_DEFER2=$d2 RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=999999 bench --steps 100 --warmup 5 2>&1 >/dev/null | grep "wait commands" | sed "s/^/defer2=$d2 /"; done
echo "== 3. abort hunt"
for i in 1 2; do
  RL_LOG_ALLOCS=1 timeout 400 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_gpu_release_lib.py > "$out/suite_$i.log" 2>&1; rc=$?
  echo "suite run $i exit: $rc"; tail -n 2 "$out/suite_$i.log" | cut -c1-200
  if [ $rc -ne 0 ]; then grep -n "Memory access fault\|terminate called\|what():\|Aborted\|HSA_STATUS\|Fatal Python" "$out/suite_$i.log" | head -20; fi
done
t_end=$(( $(date +%s) + 270 ))
n=0; bad=0
while [ $(date +%s) -lt $t_end ]; do
  n=$((n+1))
  RL_LOG_ALLOCS=1 timeout 200 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -k "multi_counter or load_counters or config5" > "$out/loop_cur.log" 2>&1; rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); cp "$out/loop_cur.log" "$out/loop_fail_$n.log"; echo "loop iteration $n exit $rc"; grep -n "Memory access fault\|terminate called\|what():\|Aborted\|HSA_STATUS\|Fatal Python\|Error" "$out/loop_cur.log" | head -20; fi
done
echo "abort hunt: $n iterations, $bad failed"; tail -n 1 "$out/loop_cur.log"
find "$out" -type f -size +4M -delete
