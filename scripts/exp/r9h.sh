#!/bin/bash
# r9h — flakiness check of what round 4 added: the wire path's tests, compaction, the fuzz sequences, three times over; the release-library test.
set -u
out=$PWD/gpurun_out/r9h; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_rls_e2e.py tests/test_gpu_fuzz.py tests/test_gpu_release_lib.py tests/test_gpu_parity.py -x -q -k "not config3 and not config2 and not one_million" -p no:cacheprovider 2>&1 | tail -3 > "$out/run$i.log"; echo "run $i exit ${PIPESTATUS[0]}: $(tail -n 1 "$out/run$i.log")"
done
