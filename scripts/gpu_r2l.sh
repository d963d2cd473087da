#!/bin/bash
set -u
mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
out=$PWD/gpurun_out/r2l
timeout 300 python scripts/bench_sweep.py > $out/sweep.json 2> $out/sweep.err; cat $out/sweep.json; tail -2 $out/sweep.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $OLDPWD/scripts/bench_sweep.py --reps 5 > $out/sweep_prof.json 2> $out/trace.err
cd $OLDPWD
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
grep -E "k_scan|k_rehash|k_table_init|Name" $f | cut -c1-150
find $out -type f -size +4M -delete
