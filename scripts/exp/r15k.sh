#!/bin/bash
# (needs scripts/exp/patches/resp_thin_copy_nocu.patch applied: the forms it measures were parked there)
# r15k — the pinned staging the responses are written into: default flags, explicitly coherent (fine-grained), explicitly non-coherent
set -u
out=$PWD/gpurun_out/r15k; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
for f in default 0x40000000 0x80000000; do
    if [ $f = default ]; then unset RL_STAGE_FLAGS; else export RL_STAGE_FLAGS=$f; fi
    timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$f.json" 2>/dev/null
    python - "$out/rls_$f.json" "staging flags $f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s, call p50 %.2f" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
