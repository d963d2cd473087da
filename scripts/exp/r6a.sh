#!/bin/bash
# r6a (prepared in round 3, not yet run): where do k_bkt_part_c's 28 us go?  Phase stamps of every tile's workgroup, alone
# (blocking calls) and beside the replay (three batches in flight; rows of a partition that is still running are skipped).
# Here, before the visit:  scripts/exp/build_variant.sh part_c_stamps
# then:                    gpurun --timeout 60 -- 'bash scripts/exp/r6a.sh'
set -u
out=$PWD/gpurun_out/r6a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cp limitador_amd/lib/variants/librl_engine_part_c_stamps.so limitador_amd/lib/librl_engine.so || exit 1   # (the box's copy of the tree)
RL_APPLY_TRACE=1 timeout 60 python bench.py --cpu-seconds 0 --secondary 0 --steps 20 --warmup 5 --depth 1 > "$out/alone.json" 2> "$out/alone.err"
grep "^\[part_c\]" "$out/alone.err" | tail -4
grep "^\[apply\]" "$out/alone.err" | tail -2
RL_APPLY_TRACE=1 timeout 60 python bench.py --cpu-seconds 0 --secondary 0 --steps 20 --warmup 5 > "$out/pipe.json" 2> "$out/pipe.err"
grep "^\[part_c\]" "$out/pipe.err" | tail -6
