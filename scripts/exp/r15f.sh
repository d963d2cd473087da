#!/bin/bash
# r15f — four serving calls in flight: how many workgroups of k_resp<true> write host memory at once (RL_RESP_WRITERS; 128 in the
# tree).  r15b / r15e: while a kernel writes host memory — k_resp or a blit copy — every kernel of the other queue runs 3-8 x longer.
set -u
out=$PWD/gpurun_out/r15f; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
for w in ${WRITERS:-128 32 16 8 4}; do
    RL_RESP_WRITERS=$w timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$w.json" 2>/dev/null
    python - "$out/rls_$w.json" "writers=$w" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s, call p50 %.2f" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
