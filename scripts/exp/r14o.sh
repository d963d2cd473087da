#!/bin/bash
# r14o — the phased round stores only failing flags too (rl_gen_round_device); the key-sharded multi-counter step at world 1
set -u
out=$PWD/gpurun_out/r14o; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
if [ -z "${SKIP_TESTS:-}" ]; then
timeout 600 python -X faulthandler -m pytest tests/test_gpu_sharded_multi.py tests/test_gpu_sharded_abi.py -q -x > "$out/sh.log" 2>&1; echo "tests exit: $?"; tail -n 2 "$out/sh.log" | cut -c1-200
fi
for cfg in 1 0 1 0; do
  echo "prefill=$cfg: $(RL_GEN_PASS_PREFILL=$cfg timeout 200 python scripts/bench_sharded_requests.py 2>/dev/null | grep '^{' | tail -n 1 | cut -c1-300)"
done
