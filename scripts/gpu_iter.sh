#!/bin/bash
# One GPU-box visit for the general resolver: bench_match, its rocprofv3 kernel stats, the k_gen_sort phase stamps.
set -u
mkdir -p gpurun_out/gen
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
out=$PWD/gpurun_out/gen
timeout 300 python scripts/bench_match.py > $out/match.json 2> $out/match.err; cut -c1-300 $out/match.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python scripts/bench_match.py --steps 6 > $out/match_prof.json 2> $out/trace.err
RL_GEN_TRACE=2 timeout 300 python scripts/bench_match.py --steps 3 2>&1 | grep "k_gen_sort:" | tail -1 > $out/sort_phases.txt; cat $out/sort_phases.txt
find $out -type f -size +4M -delete
