#!/bin/bash
set -u
out=$PWD/gpurun_out/r3r; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sharded_abi.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -4
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", d["config"]["parallelism"])
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {  # name, env..., then bench args after --
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
}
run sh_ext X=1 -- --steps 100 --warmup 5 --force-sharded
run sh_own RL_SHARDED_ENGINE_STREAMS=own -- --steps 100 --warmup 5 --force-sharded
run sh_own_lo RL_SHARDED_ENGINE_STREAMS=own RL_SHARDED_STREAM_PRIO=0 -- --steps 100 --warmup 5 --force-sharded
exit 0
cd /tmp
RL_SHARDED_ENGINE_STREAMS=own timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o t -- python $OLDPWD/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 --timing-mode 0 --force-sharded > "$out/bench_trace.json" 2> "$out/trace.err"
csv=$(find "$out/trace" -name "*kernel_trace.csv" | head -1)
python $OLDPWD/scripts/kstats.py "$csv" 20 | head -12
python - "$csv" <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])),key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-150:-90]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q{r.get('Queue_Id','?'):>3} {r['Kernel_Name'].split('(')[0][-50:]}")
PY
cd $OLDPWD
find "$out" -type f -size +6M -delete
