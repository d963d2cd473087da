#!/bin/bash
# round 3, visit b: where k_bkt_apply's time goes (phase stamps), true kernel durations (rocprofv3), tile sizes.
set -u
out=$PWD/gpurun_out/r3b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -5 > "$out/pytest.log"
tail -3 "$out/pytest.log"
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "frac", round(d["roofline"]["frac"],4), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
}
run base X=1 -- --steps 100 --warmup 5
run steps8 RL_PART_STEPS=8 -- --steps 100 --warmup 5
run steps16 RL_PART_STEPS=16 -- --steps 100 --warmup 5
run trace1 RL_APPLY_TRACE=1 -- --steps 8 --warmup 5 --depth 1 --timing-mode 0
grep "^\[apply\]" "$out/trace1.err" | tail -4
run trace3 RL_APPLY_TRACE=1 -- --steps 8 --warmup 5 --timing-mode 0
grep "^\[apply\]" "$out/trace3.err" | tail -4
run trace1u RL_APPLY_TRACE=1 -- --steps 8 --warmup 5 --depth 1 --timing-mode 0 --zipf 0
grep "^\[apply\]" "$out/trace1u.err" | tail -2
cd /tmp
for v in pipe one; do
  if [ $v = one ]; then export RL_OVERLAP=0; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$v" -o t -- python $OLDPWD/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 --timing-mode 0 > "$out/bench_trace_$v.json" 2> "$out/trace_$v.err"
  csv=$(find "$out/trace_$v" -name "*kernel_trace.csv" | head -1)
  echo "== $v"; python $OLDPWD/scripts/kstats.py "$csv" 20 | head -8
  python $OLDPWD/scripts/timeline.py "$csv" 16 > "$out/timeline_$v.txt"; head -16 "$out/timeline_$v.txt"
done
unset RL_OVERLAP
cd "$OLDPWD"
find "$out" -type f -size +6M -delete
