#!/bin/bash
# r14h — the responses' kernels go out behind the offsets' copy without the host having seen the total (RL_RESP_BLIND)
set -u
out=$PWD/gpurun_out/r14h; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py -q -x > "$out/wire.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/wire.log" | cut -c1-200
for cfg in 1 0 1 0; do
  RL_RESP_BLIND=$cfg timeout 300 python scripts/bench_rls.py hashed 4096,32768,262144 > "$out/rls.json" 2>/dev/null
  python - "$out/rls.json" "blind=$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], " ".join("%s: codes %.3f headers %.3f |"%(n, d["sizes"][n]["codes_only"]["p50_ms"], d["sizes"][n]["with_headers"]["p50_ms"]) for n in ("4096","32768","262144")))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
