#!/bin/bash
# One GPU-box visit for the GENERAL RESOLVER (k_gen_*): rocprofv3 kernel trace + stats of
# scripts/bench_match.py (1 M requests -> 3.1 M counters), then the PMC passes (HBM bytes, L2 hit rate, EA requests) each in
# its OWN run, plus the calibration kernels.  Raw output -> gpurun_out/prof_<tag>/; summaries -> profiles/ by
# scripts/summarize_gen_prof.py (run here, after the visit).
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_profile_gen.sh r06a'
set -u
tag=${1:-r06a}
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/scripts/bench_match.py --steps 6 ${GEN_ARGS:-}"
timeout 300 python scripts/bench_match.py --steps 20 ${GEN_ARGS:-} > "$out/gen_bench.json" 2> "$out/gen.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/gen" -o g -- $CMD > "$out/gen_bench_under_trace.json" 2>> "$out/gen.err"
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pmc | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- $CMD > /dev/null 2> "$out/pmc_$name.err"
  timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/calib_$name" -o c -- $REPO/scripts/microbench/bin/pmc_calib > /dev/null 2> "$out/calib_$name.err"
done
cd "$REPO"
find "$out" -type f -size +12M -delete
find "$out" -type f | head -60 > "$out/files.txt"
cat "$out/gen_bench.json"; tail -3 "$out/gen.err"
