#!/bin/bash
# r15q — FOUR serving sets (the header's constant changed in this throw-away snapshot, experiment build rebuilt on the box) with the
# transfer as copy commands (RL_RESP_VIA_COPY=1): sustained rates and the laps of the engine call ([wm]) and of the host call ([rli])
# on one clock
set -u
out=$PWD/gpurun_out/r15q; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp LIMITADOR_AMD_LIB=exp
true
python -c "from limitador_amd import build; build.build_engine(force=True); build.build_storage(force=True)" > "$out/build.log" 2>&1 || { tail -5 "$out/build.log"; exit 1; }
for v in ${VIAS:-1 0}; do export RL_RESP_PIECES=${PIECES:-8}
  RL_RESP_VIA_COPY=$v RL_WIRE_TRACE=1 RLI_TRACE=1 timeout 300 python scripts/bench_rls.py hashed 262144 > "$out/rls_$v.json" 2> "$out/laps_$v.txt"
  python - "$out/rls_$v.json" "4 sets via_copy=$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["sizes"]["262144"]
print(sys.argv[2], "one at a time: %.3f ms" % d["with_headers"]["p50_ms"], " | ".join("%s: %.3f ms, %.1f M msg/s" % (k.split("_")[2], d[k]["ms_per_batch_sustained"], d[k]["requests_per_s"]/1e6) for k in ("with_headers_two_in_flight","with_headers_three_in_flight","with_headers_four_in_flight") if k in d))
PY
  grep -E "\[wm\]|\[rli\]" "$out/laps_$v.txt" | tail -n 400 > "$out/tail_$v.txt"
done
