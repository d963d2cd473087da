#!/bin/bash
# r11d — what the replay's memory path is busy with: TA / TCP / UTCL1 (address translation) / TD counters of k_bkt_step and
# k_bkt_part_c, `rocprofv3 --pmc` passes of their own with --kernel-trace only (dispatches are serialised under PMC: the
# kernels run alone).  Two forms of the replay: without and with the read-ahead touch (one more random load per hit) — the
# counter that moves towards its ceiling with the touch is the one the rounds queue on.
set -u
out=$PWD/gpurun_out/r11d; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
root=$PWD
B="python $root/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0"
cd /tmp
i=0
for pmc in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "GRBM_GUI_ACTIVE TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for ra in 0 1; do
    RL_READ_AHEAD=$ra timeout 150 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out/p${i}_ra$ra" -o p -- $B > /dev/null 2> "$out/p${i}_ra$ra.err" || echo "pass $i ra $ra failed: $(tail -n 2 $out/p${i}_ra$ra.err | cut -c1-200)"
  done
done
cd "$root"
python - "$out" <<'PY'
import csv,glob,sys,os,collections
out=sys.argv[1]
res=collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out+"/p*_ra*")):
    if not os.path.isdir(d): continue
    ra=d[-1]
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "k_bkt_step" in k or "k_bkt_part_c" in k:
                res[(k.split("<")[0],ra)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out+"/summary.txt","w") as fo:
    for (k,ra),cs in sorted(res.items()):
        line=f"{k} read_ahead={ra}: "+", ".join(f"{c}={sum(v[-20:])/len(v[-20:]):.4g}" for c,v in sorted(cs.items()))
        print(line); fo.write(line+"\n")
PY
find "$out" -type f -size +3M -delete
