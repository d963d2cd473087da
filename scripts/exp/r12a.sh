#!/bin/bash
# r12a — FIRST VISIT OF THE NEXT ROUND (prepared at the end of round 4, when the GPU budget was spent).
# RL_DEFER2=1 (experiment build): a replay whose partition is still running when the next batch is submitted is held back
# across that submit and goes out from the collect's spin — no wait command in front of it (rl_engine::pend_old).  What it is
# after: scripts/microbench/kernel_gap2.hip (profiles/r04h_kernel_gap2.txt) — a wait on an event that is not complete when it
# is enqueued costs the stream 5.4 us at the kernel boundary (3.6 -> 9.0 us), and four replays out of five carry one.
#   1. parity with the switch on (the opt-in runs of the pipeline suites, then the full-size in-flight tests)
#   2. the bench with and without it (expected: the replay stream's idle time 9.7 -> ~4.5 us, the step 45 -> ~41 us over 200 steps)
# If green and faster: make it the default (rl_engine::defer2 = true), drop RL_TEST_DEFER2 from the two test files.
set -u
out=$PWD/gpurun_out/r12a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
RL_TEST_DEFER2=1 timeout 600 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -x -q 2>&1 | tail -6 > "$out/pytest_defer2.log"; echo "pytest (opt-in runs) exit: ${PIPESTATUS[0]}"; tail -n 3 "$out/pytest_defer2.log"
RL_DEFER2=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "config3 or config2 or in_flight or every_window or third_of" 2>&1 | tail -4 > "$out/pytest_full_size.log"; echo "pytest (full size, RL_DEFER2=1) exit: ${PIPESTATUS[0]}"; tail -n 2 "$out/pytest_full_size.log"
bench() { timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for d2 in 0 1 0 1; do
  for steps in 20 200; do
    RL_DEFER2=$d2 bench --steps $steps --warmup 6 > "$out/d$d2_s$steps.json" 2> "$out/d$d2_s$steps.err"
    python - "$out/d$d2_s$steps.json" "defer2=$d2 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "replay stream idle", round(p["apply_stream_idle_ms_per_batch"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
# RL_TIMING_LAZY=1 (also prepared without a GPU): a timed batch's events are read at the start of the NEXT collect instead of
# between its own collect and the next submit (the 4-5 us every fourth gap carries on top).  Parity is not involved (host-side
# reads of events); what to check: roofline.avg_launch_ms / pipeline.kernel_ms_per_batch are still filled and agree with rocprofv3.
for cfg in "0 0" "0 1" "1 1"; do
  set -- $cfg
  for steps in 20 200; do
    RL_DEFER2=$1 RL_TIMING_LAZY=$2 bench --steps $steps --warmup 6 > "$out/d$1_l$2_s$steps.json" 2> "$out/d$1_l$2_s$steps.err"
    python - "$out/d$1_l$2_s$steps.json" "defer2=$1 timing_lazy=$2 steps=$steps" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step launch", round(d["roofline"]["avg_launch_ms"]*1e3,1), "part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "replay stream idle", round(p["apply_stream_idle_ms_per_batch"]*1e3,1))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
# how many wait commands went in (printed when the engine is destroyed; the trace mode itself slows the host down)
for d2 in 0 1; do RL_DEFER2=$d2 RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=999999 bench --steps 100 --warmup 5 2>&1 >/dev/null | grep "wait commands" | sed "s/^/defer2=$d2 /"; done
