#!/bin/bash
# r15m — device timeline of a SMALL serving call (256 hashed-key messages, headers: responses assembled on the host)
set -u
out=$PWD/gpurun_out/r15m; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
REPO=$PWD
timeout 120 python scripts/bench_rls.py hashed 256,4096 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['sizes']; print({n:{k:round(v['p50_ms'],4) for k,v in r.items() if 'p50_ms' in v} for n,r in d.items()})"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/tr" -o t -- python $REPO/scripts/bench_rls.py hashed 256 > "$out/rls.json" 2> "$out/tr.err"
cd "$REPO"
python scripts/timeline_tail.py "$out/tr" 0.9 0 0 > "$out/timeline.txt"
cat "$out/timeline.txt" | tail -60
