#!/bin/bash
# r13c — third visit of round 5: the GPU suite on the tree with keyed hashed keys (SipHash), the all-at-once collision
# protocol and CrCounterValue-exact local restarts (f4); then the replay's stores written through (RL_STEP_WT variants)
# against the default, now that the boundary between two replays is the dirty-L2 write-back.
set -u
out=$PWD/gpurun_out/r13c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
echo "== 1. the touched suites first, then everything"
timeout 600 python -X faulthandler -m pytest tests/test_gpu_merge.py tests/test_gpu_rls_e2e.py tests/test_gpu_match.py -q -x > "$out/touched.log" 2>&1; rc=$?
echo "touched suites exit: $rc"; tail -n 3 "$out/touched.log" | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED\|Timeout\|Memory access fault\|Aborted" "$out/touched.log" | head -40; fi
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --deselect tests/test_gpu_release_lib.py::test_the_bench_configuration_at_full_size_on_the_release_build > "$out/suite.log" 2>&1; rc=$?
echo "suite exit: $rc"; grep -n "passed\|failed" "$out/suite.log" | tail -n 2 | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^FAILED\|^ERROR\|Timeout\|Memory access fault\|Aborted" "$out/suite.log" | head -40; fi
echo "== 2. write-through stores in the replay"
export LIMITADOR_AMD_LIB=exp
bench() { timeout 150 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step; replay", round(d["roofline"]["avg_launch_ms"]*1e3,1), "alone", round((d["roofline"]["avg_launch_ms_alone"] or 0)*1e3,1), "; part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "; idle", round(p["apply_stream_idle_ms_per_batch"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
cp limitador_amd/lib/exp/librl_engine.so "$out/exp_default.so"
for rep in 1 2; do
  for v in default step_wt1 step_wt2; do
    if [ $v = default ]; then cp "$out/exp_default.so" limitador_amd/lib/exp/librl_engine.so; else cp limitador_amd/lib/variants/librl_engine_$v.so limitador_amd/lib/exp/librl_engine.so; fi
    for steps in 20 200; do
      f="$out/${v}_s${steps}_r$rep.json"
      bench --steps $steps --warmup 6 > "$f" 2> "${f%.json}.err"; show "$f" "$v steps=$steps rep=$rep"
    done
  done
done
# parity of the write-through variants on the pipeline suites (they would only be adopted green)
for v in step_wt1 step_wt2; do
  cp limitador_amd/lib/variants/librl_engine_$v.so limitador_amd/lib/exp/librl_engine.so
  timeout 600 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -q -x > "$out/parity_$v.log" 2>&1; echo "$v parity exit: $?"; tail -n 1 "$out/parity_$v.log" | cut -c1-200
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "config3_through or every_window" > "$out/parity_full_$v.log" 2>&1; echo "$v full-size parity exit: $?"; tail -n 1 "$out/parity_full_$v.log" | cut -c1-200
done
cp "$out/exp_default.so" limitador_amd/lib/exp/librl_engine.so; rm -f "$out/exp_default.so"
unset LIMITADOR_AMD_LIB
echo "== 3. the driver's line (release build)"
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench exit: $?"; cut -c1-900 "$out/bench.json"
find "$out" -type f -size +4M -delete
