#!/bin/bash
# r9f — k_gen_post zeroes the scratch of an applied pass (no host fills between two calls): resolver tests + bench_match + timeline.
set -u
out=$PWD/gpurun_out/r9f; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py tests/test_gpu_fuzz.py tests/test_gpu_rls_e2e.py tests/test_gpu_host_mirror.py -x -q -k "not one_million" 2>&1 | tail -6 > "$out/pytest.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest.log"
timeout 300 python scripts/bench_match.py --steps 20 > "$out/match.json" 2> "$out/match.err"
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o t -- python $OLDPWD/scripts/bench_match.py --steps 6 > "$out/match_prof.json" 2> "$out/trace.err" )
find "$out" -type f -size +6M -delete
tail -n 4 "$out/pytest.log"; cut -c1-330 "$out/match.json"
