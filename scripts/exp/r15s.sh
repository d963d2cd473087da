#!/bin/bash
# r15s — the lazily issued transfer: pieces per call (RL_RESP_PIECES) and pieces of a set in flight at once (RL_RESP_LAZY_DEPTH);
# experiment build, four sets, 2 / 3 / 4 callers
set -u
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for cfg in "8 1" "8 2" "4 1" "16 1" "16 2" "8 1"; do
  set -- $cfg
  RL_RESP_PIECES=$1 RL_RESP_LAZY_DEPTH=$2 timeout 300 python scripts/bench_rls.py hashed 262144 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['sizes']['262144']
print('pieces $1 depth $2: one at a time %.3f ms |' % d['with_headers']['p50_ms'], ' | '.join('%s: %.3f ms %.1f M/s' % (k.split('_')[2], d[k]['ms_per_batch_sustained'], d[k]['requests_per_s']/1e6) for k in ('with_headers_two_in_flight','with_headers_three_in_flight','with_headers_four_in_flight')))"
done
