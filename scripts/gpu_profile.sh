#!/bin/bash
# One GPU-box visit: rocprofv3 kernel trace of bench.py (timing), then the PMC passes (HBM bytes,
# L2 hit rate) in their OWN runs, plus the same counters over kernels with known byte counts
# (scripts/microbench/pmc_calib.hip) to calibrate them.  Raw output -> gpurun_out/prof_<tag>/,
# summaries are copied into profiles/ by scripts/summarize_prof.py (run here, after the visit).
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_profile.sh r01b'
set -u
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0"
cd /tmp
# 1. timing: kernel trace + stats (no counters in this run)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $BENCH > "$out/bench_under_trace.json" 2> "$out/trace.err"
# 2. counters, one pass each (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2)
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pmc | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- $BENCH > /dev/null 2> "$out/pmc_$name.err"
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/calib_$name" -o c -- $OLDPWD/scripts/microbench/bin/pmc_calib > /dev/null 2> "$out/calib_$name.err"
done
# 2b. the general resolver (match + multi-counter check_and_update): kernel trace of scripts/bench_match.py
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/gen" -o g -- python $OLDPWD/scripts/bench_match.py --steps 6 > "$out/gen_bench_under_trace.json" 2> "$out/gen.err"
cd "$OLDPWD"
timeout 200 python scripts/bench_match.py --steps 20 > "$out/gen_bench.json" 2>> "$out/gen.err"
# keep the merged-back output small: per-dispatch CSVs of the engine's kernels only
find "$out" -type f -size +8M -delete
find "$out" -type f | head -50 > "$out/files.txt"
# 3. the plain bench line (with the CPU baseline) for BENCH comparison
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
tail -2 "$out/bench.err"; cat "$out/bench.json" | cut -c1-600
ls "$out"
