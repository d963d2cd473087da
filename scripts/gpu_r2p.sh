#!/bin/bash
mkdir -p gpurun_out/r2p
timeout 600 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED|Error" | head -10
for i in 1 2; do
timeout 300 python bench.py --steps 300 --warmup 10 --cpu-seconds 0 --secondary 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('value %.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'apply ovl %.1f alone %.1f'%(r['avg_launch_ms']*1e3, r['avg_launch_ms_alone']*1e3), {k:round(v*1e3,1) for k,v in d['pipeline']['kernel_ms_per_batch'].items()})"
done
