#!/bin/bash
# r14s — the whole GPU suite + smoke on the round's last tree
set -u
out=$PWD/gpurun_out/r14s; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q > "$out/pytest_gpu_full.log" 2>&1; echo "pytest exit: $?"; grep -E "passed|failed" "$out/pytest_gpu_full.log" | tail -n 2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
