#!/bin/bash
# r9d — f1 finished: the wire path with the messages decoded on the device and counters keyed by a hash of their
# canonical key bytes (rl_wire.hpp): the e2e tests in both key modes, then the latency table of both.
set -u
out=$PWD/gpurun_out/r9d; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_match.py -x -q 2>&1 | tail -30 > "$out/pytest_rls.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_rls.log"
for k in exact hashed; do timeout 300 python scripts/bench_rls.py $k > "$out/rls_$k.json" 2> "$out/rls_$k.err"; done
tail -n 30 "$out/pytest_rls.log"
python - "$out" <<'PY'
import json,sys
for k in ("exact","hashed"):
    try:
        d=json.load(open(f"{sys.argv[1]}/rls_{k}.json"))
        print(k, {n:(round(v["codes_only"]["p50_ms"],3), round(v["with_headers"]["p50_ms"],3)) for n,v in d["sizes"].items()})
    except Exception as ex: print(k,"FAILED",ex, open(f"{sys.argv[1]}/rls_{k}.err").read()[-600:])
PY
