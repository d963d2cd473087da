#!/bin/bash
# r14a — the Kuadrant methods (rl_match_batch_op / rli_serve_batch_op) on the GPU, the wire / match suites behind the
# refactor, and a fresh kernel timeline of the general resolver (scripts/bench_match.py) to aim the next cut at
set -u
out=$PWD/gpurun_out/r14a; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_kuadrant.py tests/test_gpu_rls_e2e.py tests/test_gpu_match.py -q -x > "$out/kuadrant.log" 2>&1; echo "tests exit: $?"; tail -n 15 "$out/kuadrant.log" | cut -c1-220
timeout 200 python scripts/bench_match.py --steps 20 > "$out/gen_bench.json" 2> "$out/gen_bench.err"; cut -c1-400 "$out/gen_bench.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/gen" -o g -- python $OLDPWD/scripts/bench_match.py --steps 6 > "$out/gen_under_trace.json" 2> "$out/gen.err"
f=$(find "$out/gen" -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $OLDPWD/scripts/timeline.py "$f" 30 | cut -c1-110
cd "$OLDPWD"
find "$out" -type f -size +6M -delete
