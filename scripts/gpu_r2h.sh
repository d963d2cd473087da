#!/bin/bash
set -u
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
out=$PWD/gpurun_out/r2h
B="python $PWD/bench.py --steps 12 --warmup 4 --cpu-seconds 0 --timing-mode 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace3 -o t -- $B --depth 3 > $out/trace3.json 2> $out/trace3.err
cd $OLDPWD
f=$(find $out/trace3 -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $f 40
find $out -type f -size +4M -delete
