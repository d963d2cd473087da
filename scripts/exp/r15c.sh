#!/bin/bash
# r15c — four serving sets: 2, 3 and 4 serving calls in flight (release build)
set -u
out=$PWD/gpurun_out/r15c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py -q -x --timeout 120 > "$out/tests.log" 2>&1; echo "tests exit: $?"; tail -n 2 "$out/tests.log" | cut -c1-200
for rep in 1 2 3; do
  timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$rep.json" 2>/dev/null
  python - "$out/rls_$rep.json" "rep$rep" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms per batch, %.1f M msg/s, call p50 %.2f ms" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
