#!/bin/bash
# One GPU-box visit: the GPU test suite, smoke, and the bench line.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh'
# (profiling visit: scripts/gpu_profile.sh <tag>, then scripts/summarize_prof.py here)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc > gpurun_out/host.txt; lscpu | grep -E 'Model name|^CPU\(s\)|Thread|Socket' >> gpurun_out/host.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit: $?" >> gpurun_out/bench.err
for f in gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.err; do tail -n 4 $f; done
cat gpurun_out/bench.json
