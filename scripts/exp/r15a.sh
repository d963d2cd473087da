#!/bin/bash
# r15a — two serving calls in flight: do the response bytes leave better in copy commands (RL_RESP_VIA_COPY=1: k_resp into a device
# buffer, hipMemcpyAsync per piece on the responses' stream) than as 128 workgroups' stores into pinned memory?
set -u
out=$PWD/gpurun_out/r15a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
RL_RESP_VIA_COPY=1 timeout 200 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py -q -x --timeout 120 > "$out/tests.log" 2>&1; echo "tests (via copy) exit: $?"; tail -n 2 "$out/tests.log" | cut -c1-200
for rep in 1 2 3; do
  for v in 0 1; do
    RL_RESP_VIA_COPY=$v timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$v.json" 2>/dev/null
    python - "$out/rls_$v.json" "rep$rep via_copy=$v" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    t=d.get("with_headers_two_in_flight",{})
    print(sys.argv[2], "one at a time: %.3f ms | two in flight: %.3f ms per batch, %.1f M msg/s" % (d["with_headers"]["p50_ms"], t.get("ms_per_batch_sustained",0), t.get("requests_per_s",0)/1e6))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
