#!/bin/bash
# r14g — the first blind group of fixpoint rounds follows the pass before (gen_rounds_hint): the general resolver's tests,
# the wire path and bench_match with it
set -u
out=$PWD/gpurun_out/r14g; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py tests/test_gpu_match.py tests/test_gpu_merge.py tests/test_gpu_sharded_multi.py -q -x > "$out/gen.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/gen.log" | cut -c1-200
for rep in 1 2; do
  timeout 300 python scripts/bench_rls.py hashed 32768,262144 > "$out/rls.$rep.json" 2>/dev/null
  python - "$out/rls.$rep.json" "hashed rep$rep" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for n in ("32768","262144"):
        r=d["sizes"][n]; print(sys.argv[2], n, "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"], "check %.3f report %.3f"%(r["kuadrant_check"]["p50_ms"], r["kuadrant_report"]["p50_ms"]))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
timeout 200 python scripts/bench_match.py --steps 20 | cut -c1-300
RL_GEN_TRACE=1 timeout 200 python scripts/bench_rls.py hashed 262144 2>&1 >/dev/null | grep "\[gen\]" | tail -n 4 | cut -c1-200
