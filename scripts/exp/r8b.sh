#!/bin/bash
# r8b — what is the step bound by?  The table shrunk until it lives in the Infinity Cache (1 M keys: 134 MB) and in the L2s
# (150 k keys: 16 MB; every XCD's bucket range is a contiguous eighth of the table = 2 MB of its 4 MB L2), Zipf and uniform
# keys, with and without the dependency-free overlap of consecutive replays (RL_XOVER=2: wrong results, best-case timing).
set -u
out=$PWD/gpurun_out/r8b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
bench() { timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for keys in 10000000 1000000 150000; do
 for z in 0.99 0; do
  for x in 0 2; do
    RL_XOVER=$x bench --steps 200 --warmup 10 --keys $keys --zipf $z > "$out/k${keys}_z${z}_x${x}.json" 2> "$out/k${keys}_z${z}_x${x}.err"
  done
 done
done
python - "$out"/*.json <<'PY'
import json,sys,os
for f in sys.argv[1:]:
    try:
        d=[json.loads(l) for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]; p=d["pipeline"]
        print(os.path.basename(f), round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "denied", d["config"]["denied_in_last_batch"])
    except Exception as ex: print(f,"FAILED",ex)
PY
