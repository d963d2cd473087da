#!/bin/bash
# (needs scripts/exp/patches/resp_thin_copy_nocu.patch applied: the forms it measures were parked there)
# r15o — the thin-copy response transfer as the tree's form (RL_RESP_VIA_COPY=3, 32 workgroups) against k_resp<true> writing the
# staging itself (=0): wire tests under the new default, then the bench in both forms (experiment build), sizes 4096 .. 262144
set -u
out=$PWD/gpurun_out/r15o; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
timeout 400 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py -q -x --timeout 150 > "$out/tests.log" 2>&1; echo "tests exit: $?"; tail -n 2 "$out/tests.log" | cut -c1-200
for rep in 1 2 3; do
  for v in 0 3; do
    RL_RESP_VIA_COPY=$v timeout 300 python scripts/bench_rls.py hashed 4096,32768,262144 > "$out/rls_$v.json" 2>/dev/null
    python - "$out/rls_$v.json" "rep$rep via_copy=$v" <<'PY'
import json,sys
try:
    s=json.load(open(sys.argv[1]))["sizes"]; d=s["262144"]
    print(sys.argv[2], "headers 4096: %.3f  32768: %.3f  262144: %.3f ms |" % tuple(s[k]["with_headers"]["p50_ms"] for k in ("4096","32768","262144")), " | ".join("%s: %.3f ms" % (k.split("_")[2], d[k]["ms_per_batch_sustained"]) for k in ("with_headers_two_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
