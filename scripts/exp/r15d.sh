#!/bin/bash
# r15d — 2 / 3 / 4 serving calls in flight, response bytes as 128 workgroups' stores into pinned memory (RL_RESP_VIA_COPY=0) or
# through a device buffer and copy commands (=1): the kernel trace of four in flight (r15b) shows every kernel of the decide
# phase stretched to the length of a k_resp<true> piece while one runs.
set -u
out=$PWD/gpurun_out/r15d; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
for rep in 1 2; do
  for v in 0 1; do
    RL_RESP_VIA_COPY=$v timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$v.json" 2>/dev/null
    python - "$out/rls_$v.json" "rep$rep via_copy=$v" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s, call p50 %.2f" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
