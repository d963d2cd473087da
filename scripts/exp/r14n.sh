#!/bin/bash
# r14n — k_wire_count parses its 256 messages from LDS (the workgroup's contiguous byte range staged with 16-byte loads)
set -u
out=$PWD/gpurun_out/r14n; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
timeout 120 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py -q -x -k "wire_to_the_wire and hashed" > "$out/first.log" 2>&1 || { echo "first test failed"; tail -n 5 "$out/first.log" | cut -c1-200; exit 1; }
timeout 900 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py -q -x > "$out/wire.log" 2>&1; echo "tests exit: $?"; tail -n 3 "$out/wire.log" | cut -c1-200
for rep in 1 2 3; do
  timeout 300 python scripts/bench_rls.py hashed 32768,262144 > "$out/rls.json" 2>/dev/null
  python - "$out/rls.json" "rep$rep" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], " ".join("%s: codes %.3f headers %.3f check %.3f |"%(n, d["sizes"][n]["codes_only"]["p50_ms"], d["sizes"][n]["with_headers"]["p50_ms"], d["sizes"][n]["kuadrant_check"]["p50_ms"]) for n in ("32768","262144")))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/tr" -o t -- python $OLDPWD/scripts/bench_rls.py hashed 262144 > /dev/null 2> "$out/tr.err"
f=$(find "$out/tr" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep -E "k_wire|k_resp|k_gen_round|k_gen_sort|k_gen_load" "$f" | cut -d, -f1-5 | cut -c1-150
