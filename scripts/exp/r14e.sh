#!/bin/bash
# r14e — k_resp<true> staged through LDS, the responses' copy in chunks with rl_serve_wait (the scatter overlaps it)
set -u
out=$PWD/gpurun_out/r14e; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_kuadrant.py tests/test_gpu_rls_e2e.py -q > "$out/wire.log" 2>&1; echo "tests exit: $?"; tail -n 6 "$out/wire.log" | cut -c1-220
for k in hashed exact; do
  timeout 300 python scripts/bench_rls.py $k 256,32768,262144 > "$out/rls_$k.json" 2> "$out/rls_$k.err"
  python - "$out/rls_$k.json" "$k" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for n in ("256","32768","262144"):
        r=d["sizes"][n]; print(sys.argv[2], n, "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
RLI_TRACE=1 timeout 300 python scripts/bench_rls.py hashed 262144 2>&1 >/dev/null | grep rli | tail -n 3
for k in hashed exact; do RL_RESP_DIRECT=0 timeout 300 python scripts/bench_rls.py $k 32768,262144 > "$out/rls_${k}_copy.json" 2>/dev/null; python - "$out/rls_${k}_copy.json" "$k RL_RESP_DIRECT=0" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for n in ("32768","262144"):
        r=d["sizes"][n]; print(sys.argv[2], n, "codes %.3f ms"%r["codes_only"]["p50_ms"], "headers %.3f ms"%r["with_headers"]["p50_ms"])
except Exception as ex: print(sys.argv[2], "FAILED", ex)
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/tr" -o t -- python $OLDPWD/scripts/bench_rls.py hashed 262144 > /dev/null 2> "$out/tr.err"
cd "$OLDPWD"
python - "$out/tr" <<'PY'
import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(sys.argv[1]+"/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size","?"))))
rows.sort()
idx=[i for i,r in enumerate(rows) if "k_resp<false>" in r[2]]
if idx:
    rows=rows[idx[-1]-3:]
    t0=rows[0][0]; prev=None
    for s,e,n in rows:
        print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  gap {((s-prev)/1e3 if prev else 0):8.1f}  {n}")
        prev=max(prev or 0,e)
PY
find "$out" -type f -size +6M -delete
