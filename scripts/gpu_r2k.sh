#!/bin/bash
set -u
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|Error" | head -20
timeout 300 python scripts/bench_match.py > gpurun_out/r2k/match.json 2> gpurun_out/r2k/match.err; cut -c1-400 gpurun_out/r2k/match.json; tail -2 gpurun_out/r2k/match.err
out=$PWD/gpurun_out/r2k
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $OLDPWD/scripts/bench_match.py --steps 6 > $out/match_prof.json 2> $out/trace.err
cd $OLDPWD
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
head -22 $f | cut -c1-130
find $out -type f -size +4M -delete
