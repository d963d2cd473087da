#!/bin/bash
set -u
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1; done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
