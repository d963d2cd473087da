#!/bin/bash
# r8a — what would overlapping the replays of consecutive batches buy?  RL_XOVER=2 (experiment build): the replay of odd
# batches goes to a second stream with NO dependency between consecutive replays — the results are wrong (two replays race
# on the cells), the TIMING is what a correct hand-over could at best reach.  Decides whether the per-bucket hand-over
# (VERDICT r03 next #1d) is worth building.
set -u
out=$PWD/gpurun_out/r8a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
bench() { timeout 120 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
summ() { python - "$@" <<'PY'
import json,sys,os
for f in sys.argv[1:]:
    try:
        d=[json.loads(l) for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]; p=d["pipeline"]
        print(os.path.basename(f), round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()})
    except Exception as ex: print(f,"FAILED",ex)
PY
}
for x in 0 2; do
  RL_XOVER=$x bench --steps 200 --warmup 10 > "$out/x${x}_200.json" 2> "$out/x${x}_200.err"
  RL_XOVER=$x bench --steps 20 --warmup 5 > "$out/x${x}_20.json" 2> "$out/x${x}_20.err"
  RL_XOVER=$x bench --steps 1000 --warmup 10 > "$out/x${x}_1000.json" 2> "$out/x${x}_1000.err"
done
summ "$out"/x0_200.json "$out"/x0_20.json "$out"/x0_1000.json "$out"/x2_200.json "$out"/x2_20.json "$out"/x2_1000.json
tail -3 "$out"/x2_200.err
