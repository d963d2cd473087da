#!/bin/bash
# r14i — k_gen_round stores only the FAILING pass flags (the round's flags prefilled with 1: RL_GEN_PASS_PREFILL)
set -u
out=$PWD/gpurun_out/r14i; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py tests/test_gpu_match.py tests/test_gpu_merge.py tests/test_gpu_sharded_multi.py tests/test_gpu_variants.py -q -x > "$out/gen.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/gen.log" | cut -c1-200
for cfg in 1 0 1 0; do
  echo "prefill=$cfg: $(RL_GEN_PASS_PREFILL=$cfg timeout 200 python scripts/bench_match.py --steps 20 | cut -c150-260)"
  RL_GEN_PASS_PREFILL=$cfg timeout 300 python scripts/bench_rls.py hashed 32768,262144 > "$out/rls.json" 2>/dev/null
  python - "$out/rls.json" "prefill=$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], " ".join("%s: codes %.3f headers %.3f |"%(n, d["sizes"][n]["codes_only"]["p50_ms"], d["sizes"][n]["with_headers"]["p50_ms"]) for n in ("32768","262144")))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/gen" -o g -- python $OLDPWD/scripts/bench_match.py --steps 6 > /dev/null 2> "$out/gen.err"
f=$(find "$out/gen" -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $OLDPWD/scripts/timeline.py "$f" 24 | cut -c1-110
