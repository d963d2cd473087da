#!/bin/bash
# One GPU-box visit at the end of a round: the GPU suite, smoke, the bench line (with secondary block and CPU baseline),
# then the profile of the same command: rocprofv3 kernel trace + the PMC passes (each in its own run) + calibration.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_final.sh r02d'; then python scripts/summarize_prof.py gpurun_out/prof_<tag> <tag>
set -u
tag=${1:-r02d}
mkdir -p gpurun_out
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc > gpurun_out/host.txt; lscpu | grep -E 'Model name|^CPU\(s\)|Thread|Socket' >> gpurun_out/host.txt
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full.log 2>&1; rc=$?; tail -40 gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu.log; echo "pytest exit: $rc" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit: $?" >> gpurun_out/bench.err
cp gpurun_out/bench.json "$out/bench.json"; cp gpurun_out/bench.err "$out/bench.err"
BENCH="python $PWD/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $BENCH > "$out/bench_under_trace.json" 2> "$out/trace.err"
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pmc | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- $BENCH > /dev/null 2> "$out/pmc_$name.err"
  timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/calib_$name" -o c -- $OLDPWD/scripts/microbench/bin/pmc_calib > /dev/null 2> "$out/calib_$name.err"
done
[ -n "${SKIP_SQ:-}" ] || for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  name=$(echo $pmc | tr ' ' '+')
  RL_OVERLAP=0 timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/sq_$name" -o p -- $BENCH > /dev/null 2> "$out/sq_$name.err"
done
cd "$OLDPWD"
find "$out" -type f -size +8M -delete
find "$out" -type f | head -60 > "$out/files.txt"
for f in gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.err; do tail -n 4 $f; done
cut -c1-700 gpurun_out/bench.json
