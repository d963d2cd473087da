#!/bin/bash
# Which device array of rl_engine_create is read before it is written?  The tests given, under RL_REDZONE=2 (arrays start as the
# pattern 0xA5), with only ONE array poisoned at a time (RL_REDZONE_POISON=<name>), three runs each (the failure is not every run's).
# usage: bash scripts/debug/redzone_bisect.sh "<pytest args>"
set -u
export LIMITADOR_AMD_LIB=exp TMPDIR=/tmp
args=$1
names=$(grep -o "ALLOC(e->[a-z_0-9A-Z]*" limitador_amd/csrc/rl_engine.hip | sed 's/ALLOC(//' | sort -u)
run() { timeout 300 python -m pytest $args -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1; }
echo "zones only: $(RL_REDZONE=1 run)"
echo "all poisoned: $(RL_REDZONE=2 run)"
for nm in $names; do
  f=0
  for rep in 1 2 3; do
    if RL_REDZONE=2 RL_REDZONE_POISON=$nm run | grep -q failed; then f=$((f+1)); fi
  done
  [ $f -gt 0 ] && echo "$nm: failed $f of 3"
done
echo "bisect done"
