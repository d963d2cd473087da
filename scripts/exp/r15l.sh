#!/bin/bash
# (needs scripts/exp/patches/resp_thin_copy_nocu.patch applied: the forms it measures were parked there)
# r15l — the thin copy kernel (RL_RESP_VIA_COPY=3, 16 workgroups) cut into RL_RESP_SUBPIECES launches per piece: if the decide
# phase's kernels end in the gaps between the transfer's kernels, more gaps should let the two overlap
set -u
out=$PWD/gpurun_out/r15l; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp RL_RESP_VIA_COPY=3 RL_RESP_WRITERS=16
ulimit -c 0
for sp in ${SUBS:-1 4 16 64}; do
    RL_RESP_SUBPIECES=$sp timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$sp.json" 2>/dev/null
    python - "$out/rls_$sp.json" "thin copy x $sp launches per piece" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s, call p50 %.2f" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
