#!/bin/bash
# r15n — small serving calls: results (and inputs) of a call through ONE kernel that stores into / reads the engine's pinned staging
# (ResultsOut) instead of a copy command per array (experiment build: RL_RESULTS_DIRECT=0 is the old form)
set -u
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for k in exact hashed; do
  for v in 0 1 0 1; do
    RL_RESULTS_DIRECT=$v timeout 200 python scripts/bench_rls.py $k 1,16,256,4096 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['sizes']; print('$k direct=$v', {n:{kk:round(vv['p50_ms'],4) for kk,vv in r.items() if 'p50_ms' in vv} for n,r in d.items()})"
  done
done
timeout 600 python -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_kuadrant.py tests/test_gpu_match.py -m gpu -x -q --timeout 200 2>&1 | tail -2
