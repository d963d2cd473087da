#!/bin/bash
# (needs scripts/exp/patches/resp_thin_copy_nocu.patch applied: the forms it measures were parked there)
# r15h — the responses' bytes in copy commands of kind hipMemcpyDeviceToDeviceNoCU (RL_RESP_VIA_COPY=2): does the runtime take the
# SDMA engine for them (a host -> device SDMA copy slows the resolver 5 %, the blit kernel of a device -> host copy 70-90 %)?
set -u
out=$PWD/gpurun_out/r15h; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
RL_RESP_VIA_COPY=2 timeout 200 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py -q -x --timeout 120 > "$out/tests.log" 2>&1; echo "tests (NoCU copies) exit: $?"; tail -n 2 "$out/tests.log" | cut -c1-200
for rep in 1 2; do
  for v in 0 2; do
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
