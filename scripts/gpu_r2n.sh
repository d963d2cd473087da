#!/bin/bash
mkdir -p gpurun_out/r2n
export TMPDIR=/tmp
for mode in own external own0; do
if [ $mode = own0 ]; then export RL_SHARDED_STREAM_PRIO=0; mode2=own; else mode2=$mode; fi
RL_SHARDED_ENGINE_STREAMS=$mode2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2n/prof_$mode -o abi -- python bench.py --force-sharded --sharded-impl abi --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 > gpurun_out/r2n/bench_$mode.json 2> gpurun_out/r2n/bench_$mode.err
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r2n/prof_$mode/**/abi_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("== $mode", len(rows), "kernels; columns", list(rows[0].keys()))
names = [r["Kernel_Name"] for r in rows]
# find the last 6 k_bkt_apply launches -> window
idx = [i for i, n in enumerate(names) if "k_bkt_apply" in n]
lo, hi = idx[-6], idx[-3]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo - 12:hi + 1]:
    print("%9.1f %7.1f q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
PY
done
unset RL_SHARDED_STREAM_PRIO
timeout 300 python -m pytest tests/test_gpu_sharded_abi.py -q -x 2>&1 | tail -3
for mode in own external; do
RL_SHARDED_ENGINE_STREAMS=$mode timeout 300 python bench.py --force-sharded --sharded-impl abi --steps 200 --warmup 10 --cpu-seconds 0 --secondary 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$mode', 'value %.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'])"
done
