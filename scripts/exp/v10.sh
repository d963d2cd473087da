#!/bin/bash
set -u
out=$PWD/gpurun_out/v10; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_match.py -m gpu -q -x -k "not config3" 2>&1 | tail -4 > "$out/pytest.log"
timeout 60 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --secondary 0 > "$out/bench.json" 2> "$out/bench.err"
tail -3 "$out/pytest.log"
python - "$out/bench.json" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
print(round(d["value"]/1e9,2),"G/s", round(d["ms_per_step"]*1e3,1),"us/step", {k:round(v*1e3,1) for k,v in d["pipeline"]["kernel_ms_per_batch"].items()}, "alone", round((d["roofline"].get("avg_launch_ms_alone") or 0)*1e3,1), "frac", round(d["roofline"]["frac"],4))
PY
