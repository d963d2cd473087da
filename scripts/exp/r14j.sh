#!/bin/bash
# r14j — why the wire path is slower with prefilled pass flags: rounds per pass, kernel timeline
set -u
out=$PWD/gpurun_out/r14j; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
for cfg in 1 0; do
  echo "prefill=$cfg"
  RL_GEN_PASS_PREFILL=$cfg RL_GEN_TRACE=1 timeout 200 python scripts/bench_rls.py hashed 262144 2>&1 >/dev/null | grep "\[gen\]" | awk '{print $5,$6,$7,$8,$9,$10}' | sort | uniq -c | sort -rn | head -6
done
cd /tmp
RL_GEN_PASS_PREFILL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/tr" -o t -- python $OLDPWD/scripts/bench_rls.py hashed 262144 > /dev/null 2> "$out/tr.err"
f=$(find "$out/tr" -name '*kernel_trace.csv' | head -1)
cd "$OLDPWD"
python - "$f" <<'PY'
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx=[i for i,r in enumerate(rows) if "k_wire_count" in r[2]]
# the third-last call (a codes-only one is among the last ones: print two calls)
for start in idx[-12:-10]:
    seg=rows[start:start+60]
    t0=seg[0][0]; prev=None
    for s,e,n in seg:
        if "k_wire_count" in n and s!=t0: break
        print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  gap {((s-prev)/1e3 if prev else 0):8.1f}  {n}")
        prev=max(prev or 0,e)
    print("----")
PY
