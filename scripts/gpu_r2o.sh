#!/bin/bash
set -u
mkdir -p gpurun_out/r2o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED|Error" | head -20
timeout 300 python scripts/bench_match.py > gpurun_out/r2o/match.json 2> gpurun_out/r2o/match.err; cut -c1-400 gpurun_out/r2o/match.json; tail -2 gpurun_out/r2o/match.err
out=$PWD/gpurun_out/r2o
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python scripts/bench_match.py --steps 6 > $out/match_prof.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-50s calls %4s avg %9.1f us total %8.2f ms"%(r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $out -type f -size +4M -delete
RL_GEN_TRACE=2 timeout 300 python scripts/bench_match.py --steps 3 2>&1 | grep "k_gen_sort:" | tail -2
