#!/bin/bash
set -u
out=$PWD/gpurun_out/r4h; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for v in "RL_SERVE=1" "RL_SERVE=1 RL_SERVE_LINGER_US=20" "RL_SERVE=1 RL_SERVE_LINGER_US=1000"; do echo "== $v"; env $v RL_APPLY_TRACE=1 RL_SERVE_TIMEOUT_MS=3000 timeout 120 python scripts/debug/serve_loop.py 30000 2>&1 | grep -v amdgpu.ids | tail -3; done
RL_SERVE_TIMEOUT_MS=3000 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_mirror.py tests/test_gpu_cpp_harness.py tests/test_gpu_rls_e2e.py tests/test_gpu_fuzz.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert|FAILED" $out/pytest.log | tail -8
