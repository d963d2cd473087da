#!/bin/bash
set -u
out=$PWD/gpurun_out/r3w; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_match.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python scripts/bench_rls.py > "$out/rls_latency.json" 2> "$out/rls.err"
python - "$out/rls_latency.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for n,row in d["sizes"].items():
    print(n, {k:(round(v["p50_ms"],3), round(v["requests_per_s"]/1e6,2)) for k,v in row.items()})
PY
RLI_TRACE=1 timeout 300 python - <<'PY' 2>&1 | grep "rli\]" | tail -12
import sys, os
sys.argv=["x"]
exec(open("scripts/bench_rls.py").read().split("now = 1_700_000_000_000_000")[0])
prep = g.prepare_batch(messages(32768))
now = 1_700_000_000_000_000
for _ in range(3):
    g.serve_prepared(eng, prep, now, with_headers=False); now += 1000
PY
