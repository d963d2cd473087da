#!/usr/bin/env python
"""Compile the engine with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, SGPRs, spills, LDS bytes, waves/SIMD the compiler expects).  usage: scripts/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                      os.path.join(ROOT, "limitador_amd/csrc/rl_engine.hip"), "-o", "/tmp/_res.so",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?: \[[^\]]+\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    if flt in k:
        print(f"{k[:70]:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('TotalSGPRs')} spill(v/s)={r.get('VGPRs Spill')}/{r.get('SGPRs Spill')} "
              f"lds={r.get('LDS Size [bytes/block]')} occ={r.get('Occupancy [waves/SIMD]')} scratch={r.get('ScratchSize [bytes/lane]')}")
