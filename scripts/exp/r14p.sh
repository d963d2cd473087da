#!/bin/bash
# r14p — the forms before round 5's resolver / response changes, parity-tested behind their switches
set -u
out=$PWD/gpurun_out/r14p; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
ulimit -c 0
timeout 800 python -X faulthandler -m pytest tests/test_gpu_general_variants.py -q -x --durations=5 > "$out/v.log" 2>&1; echo "tests exit: $?"; tail -n 12 "$out/v.log" | cut -c1-200
