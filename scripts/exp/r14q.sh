#!/bin/bash
# r14q — is the 20-step figure (46 us) above the 1000-step one (41 us) because of the clocks?  perf level auto vs high
set -u
out=$PWD/gpurun_out/r14q; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
b() { timeout 200 python bench.py --cpu-seconds 0 --secondary 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%s steps: %.1f us/step, replay %.1f us, part %.1f us'%(d['steps'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['pipeline']['kernel_ms_per_batch']['part']*1e3))"; }
rocm-smi --showperflevel --showclocks 2>/dev/null | grep -E "Performance Level|sclk|mclk|fclk" | head -6
echo "-- auto"; b --steps 20 --warmup 5; b --steps 20 --warmup 5; b --steps 200 --warmup 5; b --steps 20 --warmup 100
rocm-smi --setperflevel high 2>&1 | tail -n 2
rocm-smi --showperflevel --showclocks 2>/dev/null | grep -E "Performance Level|sclk|mclk" | head -4
echo "-- high"; b --steps 20 --warmup 5; b --steps 20 --warmup 5; b --steps 200 --warmup 5
rocm-smi --setperflevel auto 2>&1 | tail -n 1
