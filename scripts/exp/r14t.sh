#!/bin/bash
# r14t — buckets of the general resolver's partition again, now that k_gen_sort does not gather (RL_GEN_BUCKET_LOG2)
set -u
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for lg in 9 10 11; do
  echo "big passes: 2^$lg buckets: $(RL_GEN_BUCKET_LOG2_BIG=$lg timeout 200 python scripts/bench_match.py --steps 20 | cut -c150-215)"
done
