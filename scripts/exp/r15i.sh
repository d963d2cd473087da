#!/bin/bash
# (needs scripts/exp/patches/resp_thin_copy_nocu.patch applied: the forms it measures were parked there)
# r15i — the responses' bytes: k_resp into a device buffer at full width, then a thin streaming copy kernel of RL_RESP_WRITERS
# workgroups to the pinned staging (RL_RESP_VIA_COPY=3), 2 / 3 / 4 calls in flight
set -u
out=$PWD/gpurun_out/r15i; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
RL_RESP_VIA_COPY=3 RL_RESP_WRITERS=16 timeout 200 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py -q -x --timeout 120 > "$out/tests.log" 2>&1; echo "tests (thin copy kernel) exit: $?"; tail -n 2 "$out/tests.log" | cut -c1-200
for w in ${WRITERS:-8 16 32 64 128}; do
    RL_RESP_VIA_COPY=3 RL_RESP_WRITERS=$w timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$w.json" 2>/dev/null
    python - "$out/rls_$w.json" "thin copy, $w workgroups" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s, call p50 %.2f" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
