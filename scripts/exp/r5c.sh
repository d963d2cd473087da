#!/bin/bash
# r5c: do five workgroups of the replay AND a partition workgroup fit a CU's registers?  A SIMD has 512 VGPRs: five replay
# waves at 80 + two partition waves at 82 (88 allocated) = 576.  Register budgets that fit: replay 72 + partition 72 (504),
# replay 72 + partition 64 with the records re-read (488).  Controls: replay 72 alone, partition 64 alone.
# The best one (by the 200-step figure) then runs the GPU suite and smoke.
set -u
out=$PWD/gpurun_out/r5c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
    p=d["pipeline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", "in-pipe", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()}, "idle", round((p.get("apply_stream_idle_ms_per_batch") or 0)*1e3,1))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 60 python bench.py --cpu-seconds 0 --secondary 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  summ "$out/$name.json"
}
for rep in 1 2; do
run A_80_82_$rep RL_X=0 -- --steps 200 --warmup 10
run B_72_72_$rep RL_APPLY2_CFG=4 RL_PART_VGPR=72 -- --steps 200 --warmup 10
run C_72_64r_$rep RL_APPLY2_CFG=4 RL_PART_VGPR=56 -- --steps 200 --warmup 10
run D_72_82_$rep RL_APPLY2_CFG=4 -- --steps 200 --warmup 10
run E_80_64r_$rep RL_PART_VGPR=56 -- --steps 200 --warmup 10
done
best=$(python - "$out" <<'PY'
import json,sys,glob,os
res={}
for f in glob.glob(sys.argv[1]+"/*_[12].json"):
    try:
        d=[json.loads(l) for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]
    except Exception: continue
    k=os.path.basename(f)[0]
    res.setdefault(k,[]).append(d["ms_per_step"])
best=min(res,key=lambda k: sum(res[k])/len(res[k])) if res else "A"
# a variant must beat the default by 2 % to be worth anything
if best!="A" and "A" in res and sum(res[best])/len(res[best]) > 0.98*sum(res["A"])/len(res["A"]): best="A"
print(best)
PY
)
echo "best: $best"
case $best in
  B) export RL_APPLY2_CFG=4 RL_PART_VGPR=72;;
  C) export RL_APPLY2_CFG=4 RL_PART_VGPR=56;;
  D) export RL_APPLY2_CFG=4;;
  E) export RL_PART_VGPR=56;;
esac
env | grep "^RL_" > "$out/suite_env.txt"
run best_20 RL_X=0 -- --steps 20 --warmup 5
timeout 80 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > "$out/pytest_gpu.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke exit: $?" >> "$out/smoke.log"
tail -3 "$out/pytest_gpu.log"; tail -2 "$out/smoke.log"
