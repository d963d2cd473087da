#!/bin/bash
# The evidence visit of round 6: everything the bench line quotes, regenerated on ONE tree.
#   GPU suite + smoke + the bench line (secondary + CPU baseline) + rocprofv3 kernel trace of the same command + the PMC passes
#   (HBM bytes, L2 hit rate, EA requests; calibration kernels) + SQ counters (in the pipeline and alone) + the TCP -> TCC request
#   counts the random-line model uses + the random-access slope microbenchmark + the general resolver's trace and PMC passes.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_final_r06.sh r06z'
#        then: python scripts/summarize_prof.py gpurun_out/prof_<tag> <tag>; python scripts/summarize_sq.py gpurun_out/prof_<tag> <tag>;
#              python scripts/summarize_gen_prof.py gpurun_out/prof_<tag>g <tag>g
set -u
tag=${1:-r06z}
mkdir -p gpurun_out
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
REPO=$PWD
rocm-smi --showproductname 2>/dev/null | head -8 > "$out/gpu.txt"
nproc > "$out/host.txt"; lscpu | grep -E 'Model name|^CPU\(s\)|Thread|Socket' >> "$out/host.txt"
git -C "$REPO" rev-parse --short HEAD > "$out/commit.txt" 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > "$out/pytest_gpu_full.log" 2>&1; rc=$?
tail -5 "$out/pytest_gpu_full.log" > "$out/pytest_gpu.txt"; echo "pytest exit: $rc" >> "$out/pytest_gpu.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke exit: $?" >> "$out/smoke.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench exit: $?" >> "$out/bench.err"
timeout 120 scripts/microbench/bin/random_slope > "$out/random_slope.txt" 2>&1
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $BENCH > "$out/bench_under_trace.json" 2> "$out/trace.err"
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pmc | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- $BENCH > /dev/null 2> "$out/pmc_$name.err"
  timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/calib_$name" -o c -- $REPO/scripts/microbench/bin/pmc_calib > /dev/null 2> "$out/calib_$name.err"
done
# the replay's requests to the L2 (what the random-line model counts): reads and writes of the vector L1s
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d "$out/tcp_req" -o p -- $BENCH > /dev/null 2> "$out/tcp_req.err"
# SQ counters, in the pipeline and alone (--depth 1)
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $pmc | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/sq_pipe_$name" -o p -- $BENCH > /dev/null 2> "$out/sq_pipe_$name.err"
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/sq_alone_$name" -o p -- $BENCH --depth 1 > /dev/null 2> "$out/sq_alone_$name.err"
done
cd "$REPO"
find "$out" -type f -size +8M -delete
find "$out" -type f | head -80 > "$out/files.txt"
# the general resolver: trace + PMC passes (scripts/gpu_profile_gen.sh) -> gpurun_out/prof_<tag>g
bash scripts/gpu_profile_gen.sh "${tag}g" > "$out/gen_visit.log" 2>&1
cat "$out/pytest_gpu.txt"; tail -2 "$out/smoke.log"; tail -2 "$out/bench.err"; cut -c1-400 "$out/bench.json"; tail -3 "$out/random_slope.txt"
