#!/bin/bash
# r9e — the whole GPU suite at the tree with the hashed wire path, the 1 M-message comparison of both key modes included.
set -u
out=$PWD/gpurun_out/r9e; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > "$out/pytest_gpu.log"; echo "pytest exit: ${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"
tail -n 12 "$out/pytest_gpu.log"
