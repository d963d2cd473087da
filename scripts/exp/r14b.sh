#!/bin/bash
# r14b — the Kuadrant methods again, and RateLimitResponse bytes built on the device (rl_resp.hpp): parity against the host
# assembly, the wire suites, then the wire path's latency per batch size in both key modes with the laps of one large batch
set -u
out=$PWD/gpurun_out/r14b; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_kuadrant.py tests/test_gpu_rls_e2e.py tests/test_gpu_match.py -q > "$out/wire.log" 2>&1; echo "tests exit: $?"; tail -n 12 "$out/wire.log" | cut -c1-220
for k in hashed exact; do
  timeout 300 python scripts/bench_rls.py $k > "$out/rls_$k.json" 2> "$out/rls_$k.err"
  python - "$out/rls_$k.json" $k <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for n in ("256","32768","262144"):
        r=d["sizes"][n]; print(sys.argv[2], n, "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  RLI_RESP_HOST=1 timeout 300 python scripts/bench_rls.py $k > "$out/rls_${k}_hostresp.json" 2> "$out/rls_${k}_hostresp.err"
  python - "$out/rls_${k}_hostresp.json" "$k(host assembly)" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for n in ("32768","262144"):
        r=d["sizes"][n]; print(sys.argv[2], n, "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
RLI_TRACE=1 timeout 300 python scripts/bench_rls.py hashed 2>&1 >/dev/null | tail -n 16
