#!/bin/bash
# r14r — the first replay after an idle stretch is not held back (a wait command on an idle stream costs nothing)
set -u
out=$PWD/gpurun_out/r14r; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
LIMITADOR_AMD_LIB=exp timeout 800 python -X faulthandler -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py tests/test_gpu_bench_config.py tests/test_gpu_release_lib.py -q -x > "$out/t.log" 2>&1; echo "tests exit: $?"; tail -n 2 "$out/t.log" | cut -c1-200
b() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%s steps: %.2f us/step, replay %.1f us, part %.1f us, idle %.1f'%(d['steps'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['pipeline']['kernel_ms_per_batch']['part']*1e3, d['pipeline']['apply_stream_idle_ms_per_batch']*1e3))"; }
for i in 1 2 3 4; do b --steps 20 --warmup 5; done
b --steps 200 --warmup 5
