#!/bin/bash
# r14f — how many pieces the direct-to-host k_resp<true> is launched in (RL_RESP_PIECES), against the copy form
set -u
out=$PWD/gpurun_out/r14f; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
run() { # label, env...
  label=$1; shift
  for rep in 1 2; do
    env "$@" timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/$label.$rep.json" 2>/dev/null
    python - "$out/$label.$rep.json" "$label rep$rep" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["sizes"]["262144"]
    print(sys.argv[2], "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
  done
}
run p8w128 RL_RESP_PIECES=8 RL_RESP_WRITERS=128
run p8w64 RL_RESP_PIECES=8 RL_RESP_WRITERS=64
run p8w32 RL_RESP_PIECES=8 RL_RESP_WRITERS=32
run p8w16 RL_RESP_PIECES=8 RL_RESP_WRITERS=16
run p16w64 RL_RESP_PIECES=16 RL_RESP_WRITERS=64
run p4w64 RL_RESP_PIECES=4 RL_RESP_WRITERS=64
run p4w32 RL_RESP_PIECES=4 RL_RESP_WRITERS=32
run copy RL_RESP_DIRECT=0
timeout 300 python -m pytest tests/test_gpu_rls_e2e.py -q -k "responses_built or wire_to_the_wire" 2>&1 | tail -n 2
