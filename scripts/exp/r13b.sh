#!/bin/bash
# r13b — second visit of round 5: the whole GPU suite on the tree with RL_DEFER2 as the default and the new tests (lean
# partition, routed sweeps, world 8 in-process, one process per rank, the RCCL warm-up), then k_bkt_part_l against
# k_bkt_part_c in the bench (experiment build: RL_PART_COMPACT=1 / 2; register budgets 96 / 128 / 80 as variant libraries).
set -u
out=$PWD/gpurun_out/r13b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
echo "== 1. GPU suite"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_gpu_release_lib.py::test_the_bench_configuration_at_full_size_on_the_release_build > "$out/suite.log" 2>&1; rc=$?
echo "suite exit: $rc"; tail -n 3 "$out/suite.log" | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert\|FAILED\|Timeout\|Memory access fault\|Aborted" "$out/suite.log" | head -40; fi
echo "== 2. lean partition A/B"
export LIMITADOR_AMD_LIB=exp
bench() { timeout 150 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]; p=d["pipeline"]
    print(sys.argv[2], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step; replay", round(d["roofline"]["avg_launch_ms"]*1e3,1), "alone", round((d["roofline"]["avg_launch_ms_alone"] or 0)*1e3,1), "; part", round(p["kernel_ms_per_batch_in_pipeline"]["part"]*1e3,1), "alone", round(p["kernel_ms_per_batch_alone"]["part"]*1e3,1), "; idle", round(p["apply_stream_idle_ms_per_batch"]*1e3,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
cp limitador_amd/lib/exp/librl_engine.so "$out/exp_default.so"
run_ab() {  # label, RL_PART_COMPACT
  for steps in 20 200; do
    f="$out/$1_s$steps.json"
    RL_PART_COMPACT=$2 bench --steps $steps --warmup 6 > "$f" 2> "${f%.json}.err"
    show "$f" "$1 steps=$steps"
  done
}
for rep in 1 2; do
  run_ab "compact_r$rep" 1
  run_ab "lean96_r$rep" 2
done
for v in partl_wpe4 partl_wpe6; do
  cp limitador_amd/lib/variants/librl_engine_$v.so limitador_amd/lib/exp/librl_engine.so
  run_ab "lean_$v" 2
done
cp "$out/exp_default.so" limitador_amd/lib/exp/librl_engine.so
rm -f "$out/exp_default.so"
# uniform keys (no hot set): the lean kernel's other regime
for pc in 1 2; do
  f="$out/uniform_pc$pc.json"
  RL_PART_COMPACT=$pc bench --steps 100 --warmup 6 --zipf 0 > "$f" 2> "${f%.json}.err"; show "$f" "uniform part_compact=$pc steps=100"
done
find "$out" -type f -size +4M -delete
