#!/bin/bash
# r14k — is the wire path's run-to-run spread (2.0 vs 2.5 ms codes-only per 262144 messages) the NUMA node the process lands on?
set -u
out=$PWD/gpurun_out/r14k; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for n in /sys/devices/system/node/node*; do echo "$(basename $n): $(cat $n/cpulist)"; done
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null)"; done | head -4
which taskset numactl 2>&1 | head -2
one() { # label, prefix...
  label=$1; shift
  "$@" timeout 200 python scripts/bench_rls.py hashed 262144 > "$out/x.json" 2>/dev/null
  python - "$out/x.json" "$label" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["sizes"]["262144"]
    print(sys.argv[2], "codes %.3f headers %.3f check %.3f report %.3f"%(r["codes_only"]["p50_ms"], r["with_headers"]["p50_ms"], r["kuadrant_check"]["p50_ms"], r["kuadrant_report"]["p50_ms"]))
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
}
for i in 1 2 3 4; do one "free$i" env; done
n0=$(cat /sys/devices/system/node/node0/cpulist); n1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
for i in 1 2; do one "node0-$i" taskset -c "$n0"; done
[ -n "$n1" ] && for i in 1 2; do one "node1-$i" taskset -c "$n1"; done
