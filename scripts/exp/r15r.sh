#!/bin/bash
# r15r — the transfer's copy commands issued lazily, one piece in flight per set (RL_RESP_VIA_COPY=4), four sets: tests first, then
# the rates with 8 / 16 / 32 pieces
set -u
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
RL_RESP_VIA_COPY=4 timeout 300 python -X faulthandler -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py -q -x --timeout 150 2>&1 | tail -2
for p in 8 16 32; do PIECES=$p VIAS="4" bash scripts/exp/r15q.sh | sed "s/^/pieces $p: /"; done
VIAS="1" bash scripts/exp/r15q.sh
