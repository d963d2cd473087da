#!/bin/bash
set -u
out=$PWD/gpurun_out/r3i; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -3
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "host", round(p.get("host_submit_us_per_batch") or 0,1), "denied", d["config"]["denied_in_last_batch"])
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"; grep "^\[engine\]" "$out/$name.err" | tail -1
}
run fused X=1 -- --steps 100 --warmup 5
run uniform X=1 -- --steps 100 --warmup 5 --zipf 0
RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_d3.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 > "$out/d3.json" 2> "$out/d3.err"
python scripts/apply_trace.py $out/trace_d3.bin | head -16
RL_APPLY_TRACE=1 RL_APPLY_TRACE_FILE=$out/trace_u3.bin timeout 120 python bench.py --cpu-seconds 0 --secondary 0 --steps 8 --warmup 5 --timing-mode 0 --zipf 0 > "$out/u3.json" 2> "$out/u3.err"
python scripts/apply_trace.py $out/trace_u3.bin | head -10
