#!/bin/bash
# r7a — first GPU visit of round 4 (VERDICT r03 "Next round" #1 a-c): evidence before any change.
#   1. scripts/microbench/bin/random_slope: the chip's random-transaction rate as a slope over 1 M .. 64 M accesses
#   2. r6c: match_digit_lean against the tree (200-step bench each, then 20 steps)
#   3. r6a: phase stamps of k_bkt_part_c (alone / in the pipeline)
#   4. r6b: stamps inside a replay round (alone / in the pipeline + raw dump of one steady-state batch)
#   5. SQ counters of k_bkt_step / k_bkt_part_c in the pipeline and alone (own rocprofv3 --pmc runs, kernel trace only)
# Before the visit: build scripts/microbench/bin/random_slope and the three variants (scripts/exp/build_variant.sh).
set -u
out=$PWD/gpurun_out/r7a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
cp limitador_amd/lib/librl_engine.so /tmp/librl_engine_tree.so
restore() { cp /tmp/librl_engine_tree.so "$root/limitador_amd/lib/librl_engine.so"; }
bench() { timeout 90 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
summ() { python - "$@" <<'PY'
import json,sys,os
for f in sys.argv[1:]:
    try:
        d=[json.loads(l) for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]; p=d["pipeline"]
        print(os.path.basename(f), round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()})
    except Exception as ex: print(f,"FAILED",ex)
PY
}
echo "== 1. random_slope"; timeout 120 scripts/microbench/bin/random_slope > "$out/random_slope.txt" 2>&1; cat "$out/random_slope.txt"

echo "== 2. match_digit_lean"
bench --steps 200 --warmup 10 > "$out/base_200.json" 2> "$out/base_200.err"
bench --steps 20 --warmup 5 > "$out/base_20.json" 2> "$out/base_20.err"
cp limitador_amd/lib/variants/librl_engine_match_digit_lean.so limitador_amd/lib/librl_engine.so
bench --steps 200 --warmup 10 > "$out/lean_200.json" 2> "$out/lean_200.err"
bench --steps 20 --warmup 5 > "$out/lean_20.json" 2> "$out/lean_20.err"
summ "$out"/base_200.json "$out"/base_20.json "$out"/lean_200.json "$out"/lean_20.json

echo "== 3. part_c stamps"
cp limitador_amd/lib/variants/librl_engine_part_c_stamps.so limitador_amd/lib/librl_engine.so
RL_APPLY_TRACE=1 bench --steps 20 --warmup 5 --depth 1 > "$out/partc_alone.json" 2> "$out/partc_alone.err"
grep "^\[part_c\]" "$out/partc_alone.err" | tail -3
RL_APPLY_TRACE=1 bench --steps 20 --warmup 5 > "$out/partc_pipe.json" 2> "$out/partc_pipe.err"
grep "^\[part_c\]" "$out/partc_pipe.err" | tail -4

echo "== 4. replay round stamps"
cp limitador_amd/lib/variants/librl_engine_apply_round_stamps.so limitador_amd/lib/librl_engine.so
RL_APPLY_TRACE=1 bench --steps 20 --warmup 5 --depth 1 > "$out/round_alone.json" 2> "$out/round_alone.err"
grep "^\[round\]" "$out/round_alone.err" | tail -3
grep "^\[apply\]" "$out/round_alone.err" | tail -2
RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=60 RL_APPLY_TRACE_FILE=$out/pipe_trace.bin bench --steps 100 --warmup 5 > "$out/round_pipe.json" 2> "$out/round_pipe.err"
grep "^\[round\]\|^\[apply\]" "$out/round_pipe.err" | tail -4
python scripts/apply_trace.py "$out/pipe_trace.bin" > "$out/pipe_trace.txt" 2>&1; head -40 "$out/pipe_trace.txt"
restore

echo "== 5. SQ counters"
B="python $root/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0"
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/sq_pipe_$i" -o p -- $B > /dev/null 2> "$out/sq_pipe_$i.err"
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/sq_alone_$i" -o p -- $B --depth 1 > /dev/null 2> "$out/sq_alone_$i.err"
done
cd "$root"
find "$out" -type f -size +6M -delete
find "$out" -name "*counter_collection.csv" | head; ls "$out"
