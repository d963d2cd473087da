#!/bin/bash
# r14l — remaining / expires_in stored once behind the rounds (k_gen_load, RL_GEN_LOAD_DEFERRED) instead of in every round
set -u
out=$PWD/gpurun_out/r14l; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py tests/test_gpu_match.py tests/test_gpu_merge.py tests/test_gpu_sharded_multi.py tests/test_gpu_sharded_abi.py tests/test_gpu_host_mirror.py -q -x > "$out/gen.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/gen.log" | cut -c1-200
for cfg in 1 0 1 0; do
  RL_GEN_LOAD_DEFERRED=$cfg timeout 300 python scripts/bench_rls.py hashed 32768,262144 > "$out/rls.json" 2>/dev/null
  python - "$out/rls.json" "deferred=$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], " ".join("%s: codes %.3f headers %.3f (+%.3f) |"%(n, d["sizes"][n]["codes_only"]["p50_ms"], d["sizes"][n]["with_headers"]["p50_ms"], d["sizes"][n]["with_headers"]["p50_ms"]-d["sizes"][n]["codes_only"]["p50_ms"]) for n in ("32768","262144")))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
