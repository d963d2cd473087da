#!/bin/bash
# r15j — is the cap of ~2.2 ms per batch with 2 / 3 / 4 calls in flight the HOST's (the hand-over scatters 39 MB of responses into
# the callers' slots)?  The callers' stride (1024 -> 256 bytes per slot) and the helper threads per call (RLI_THREADS).
set -u
out=$PWD/gpurun_out/r15j; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
run() {
    timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls.json" 2>/dev/null
    python - "$out/rls.json" "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["sizes"]["262144"]
    print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s, call p50 %.2f" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6, d[k]["call_p50_ms"]) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
run "stride 1024, threads auto"
BENCH_RLS_STRIDE=256 run "stride 256, threads auto"
BENCH_RLS_STRIDE=256 RLI_THREADS=64 run "stride 256, 64 threads"
BENCH_RLS_STRIDE=256 RLI_THREADS=16 run "stride 256, 16 threads"
RLI_THREADS=8 run "stride 1024, 8 threads"
numactl --hardware 2>/dev/null | head -12
