#!/bin/bash
set -u
OUT=gpurun_out/r3_04
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_edit_distance.py tests/test_gpu_assemble.py -x -q > "$OUT/pytest_ed.log" 2>&1 || { tail -40 "$OUT/pytest_ed.log"; exit 1; }
tail -3 "$OUT/pytest_ed.log"
timeout -k 10 500 python bench.py --cpu-sample-reads 0 --no-e2e --no-tiled --steps 5 --warmup 2 > "$OUT/bench_ed.json" 2> "$OUT/bench_ed.err" || { tail -20 "$OUT/bench_ed.err"; exit 1; }
python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r3_04/bench_ed.json') if x.startswith('{')][-1])
print(json.dumps(l['assemble_path']['validate'], indent=1))
PY
