#!/bin/bash
# r6c (prepared in round 3, not yet run): match_digit without a test or a select per bit.  Static instruction counts of
# k_bkt_part_c<8> (hipcc -S, whole kernel, eight unrolled steps): 2362 -> 1992 vector, 2186 -> 1532 scalar, 241 -> 40
# s_nop, 413 -> 222 branches; same result bit for bit (every bin id is below 2^12).  It is used by every partition kernel
# and by k_gen_sort, so the variant first runs the bench, then the GPU tests that partition.
# Here, before the visit:  scripts/exp/build_variant.sh match_digit_lean
# then:                    gpurun --timeout 200 -- 'bash scripts/exp/r6c.sh'
set -u
out=$PWD/gpurun_out/r6c; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
bench() { timeout 60 python bench.py --cpu-seconds 0 --secondary 0 "$@"; }
for rep in 1 2; do bench --steps 200 --warmup 10 > "$out/base_$rep.json" 2> "$out/base_$rep.err"; done
cp limitador_amd/lib/variants/librl_engine_match_digit_lean.so limitador_amd/lib/librl_engine.so || exit 1   # (the box's copy of the tree)
for rep in 1 2; do bench --steps 200 --warmup 10 > "$out/lean_$rep.json" 2> "$out/lean_$rep.err"; done
bench --steps 20 --warmup 5 > "$out/lean_20.json" 2> "$out/lean_20.err"
python - "$out" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=[json.loads(l) for l in open(f).read().strip().splitlines() if l.startswith("{")][-1]; p=d["pipeline"]
        print(os.path.basename(f), round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_in_pipeline"].items()}, "alone", {k:round(v*1e3,1) for k,v in p["kernel_ms_per_batch_alone"].items()})
    except Exception as ex: print(f,"FAILED",ex)
PY
timeout 150 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_match.py -x -q 2>&1 | tail -4 > "$out/pytest.log"; tail -3 "$out/pytest.log"
