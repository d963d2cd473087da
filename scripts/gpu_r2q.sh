#!/bin/bash
for b in 500000 650000 800000 1000000; do
RL_OVERLAP=0 timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --cpu-seconds 0 --secondary 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('batch $b no-overlap value %.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'apply %.1f us = %.1f ps/hit'%(r['avg_launch_ms']*1e3, r['avg_launch_ms']*1e9/$b), {k:round(v*1e3,1) for k,v in d['pipeline']['kernel_ms_per_batch'].items()})"
done
