#!/bin/bash
# r14m — the request of every hit travels with its partitioned record (b_req, RL_GEN_CARRY_REQ) instead of being gathered by k_gen_sort
set -u
out=$PWD/gpurun_out/r14m; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py tests/test_gpu_match.py tests/test_gpu_merge.py tests/test_gpu_sharded_multi.py tests/test_gpu_sharded_abi.py tests/test_gpu_host_mirror.py tests/test_gpu_variants.py -q -x > "$out/gen.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/gen.log" | cut -c1-200
for cfg in 1 0 1 0; do
  echo "carry=$cfg: $(RL_GEN_CARRY_REQ=$cfg timeout 200 python scripts/bench_match.py --steps 20 | cut -c150-260)"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/gen" -o g -- python $OLDPWD/scripts/bench_match.py --steps 6 > /dev/null 2> "$out/gen.err"
f=$(find "$out/gen" -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $OLDPWD/scripts/timeline.py "$f" 24 | cut -c1-110
