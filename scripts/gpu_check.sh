#!/bin/bash
# One GPU-box visit: parity tests, smoke, a short bench, a rocprofv3 kernel trace.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [tests|bench|prof|all]'
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc > gpurun_out/host.txt; lscpu | grep -E 'Model name|^CPU\(s\)|Thread|Socket' >> gpurun_out/host.txt
if [[ $what == all || $what == tests ]]; then
  timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
  echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit: $?" >> gpurun_out/bench.err
fi
if [[ $what == all || $what == prof ]]; then
  rm -rf gpurun_out/prof
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof -o r01 -- \
      python $OLDPWD/bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OLDPWD/gpurun_out/prof_bench.json 2> $OLDPWD/gpurun_out/prof.err )
  find gpurun_out/prof -type f > gpurun_out/prof_files.txt
  # keep only the small summaries
  find gpurun_out/prof -type f ! -name '*stats*' -size +4M -delete
fi
tail -5 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.err 2>/dev/null
cat gpurun_out/bench.json 2>/dev/null
if [[ $what == sharded ]]; then
  timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_sharded.log
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --force-sharded > gpurun_out/bench_sharded1.json 2> gpurun_out/bench_sharded1.err
  tail -5 gpurun_out/pytest_sharded.log; tail -3 gpurun_out/bench_sharded1.err; cat gpurun_out/bench_sharded1.json
fi
