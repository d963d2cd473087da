#!/bin/bash
set -u
mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "u64" 2>&1 | grep -E "^E  |passed|failed|^FAILED|Error" | head
timeout 300 python scripts/bench_match.py > gpurun_out/r2j/match.json 2> gpurun_out/r2j/match.err; cat gpurun_out/r2j/match.json; tail -2 gpurun_out/r2j/match.err
out=$PWD/gpurun_out/r2j
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $OLDPWD/scripts/bench_match.py --steps 6 > $out/match_prof.json 2> $out/trace.err
cd $OLDPWD
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-150
find $out -type f -size +4M -delete
