#!/bin/bash
# r6b (prepared in round 3, not yet run): what a 256-hit round of the replay is made of — stamps inside the first two
# rounds of every bucket workgroup (requests / phase A / phase B = the cells arrive / C / D / rebuild / second round), alone
# and beside the partition.
# Here, before the visit:  scripts/exp/build_variant.sh apply_round_stamps
# then:                    gpurun --timeout 60 -- 'bash scripts/exp/r6b.sh'
set -u
out=$PWD/gpurun_out/r6b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cp limitador_amd/lib/variants/librl_engine_apply_round_stamps.so limitador_amd/lib/librl_engine.so || exit 1   # (the box's copy of the tree)
RL_APPLY_TRACE=1 timeout 60 python bench.py --cpu-seconds 0 --secondary 0 --steps 20 --warmup 5 --depth 1 > "$out/alone.json" 2> "$out/alone.err"
grep "^\[round\]" "$out/alone.err" | tail -4
grep "^\[apply\]" "$out/alone.err" | tail -2
# beside the partition: one steady-state batch of a three-deep pipeline (every batch in flight has its own stamp buffer;
# RL_APPLY_TRACE_AT looks at one batch only, so the host is not held back by a copy per collect).  The raw dump answers
# "which workgroups start late, which end last" (scripts/apply_trace.py).
RL_APPLY_TRACE=1 RL_APPLY_TRACE_AT=60 RL_APPLY_TRACE_FILE=$out/pipe_trace.bin timeout 60 python bench.py --cpu-seconds 0 --secondary 0 --steps 100 --warmup 5 > "$out/pipe.json" 2> "$out/pipe.err"
grep "^\[round\]\|^\[apply\]" "$out/pipe.err" | tail -4
python scripts/apply_trace.py "$out/pipe_trace.bin" > "$out/pipe_trace.txt" 2>&1; head -30 "$out/pipe_trace.txt"
