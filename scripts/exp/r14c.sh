#!/bin/bash
# r14c — the laps of one large batch of the wire path (experiment build: RLI_TRACE), device responses vs host assembly
set -u
out=$PWD/gpurun_out/r14c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 300 python -X faulthandler -m pytest tests/test_gpu_kuadrant.py -q > "$out/k.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/k.log" | cut -c1-200
for k in hashed exact; do
  for h in 0 1; do
    if [ $h = 1 ]; then export RLI_RESP_HOST=1; else unset RLI_RESP_HOST; fi
    timeout 300 python scripts/bench_rls.py $k > "$out/rls_${k}_host$h.json" 2> "$out/rls_${k}_host$h.err"
    python - "$out/rls_${k}_host$h.json" "$k host_assembly=$h" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for n in ("32768","262144"):
        r=d["sizes"][n]; print(sys.argv[2], n, "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
done
unset RLI_RESP_HOST
RLI_TRACE=1 timeout 300 python scripts/bench_rls.py hashed 2>&1 >/dev/null | grep rli | tail -n 12
RLI_TRACE=1 RLI_RESP_HOST=1 timeout 300 python scripts/bench_rls.py hashed 2>&1 >/dev/null | grep rli | tail -n 6
