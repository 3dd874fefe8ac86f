#!/bin/bash
# the exchange's wire form on the GPU box: its tests, the N > 1 step at world 1 over RCCL with either format (A/B/A/B), a fuzz run
set -e
OUT=gpurun_out/r3_04
mkdir -p "$OUT"
python -m pytest tests/test_gpu_parity.py -x -q -k "wire or bench or shards" > "$OUT/pytest_wire.log" 2>&1 || { tail -40 "$OUT/pytest_wire.log"; exit 1; }
tail -3 "$OUT/pytest_wire.log"
for i in 1 2; do
for f in whole wire; do
python bench.py --force-dist --kernels-only --scaling weak --steps 20 --warmup 3 --exchange-format $f > "$OUT/weak_w1_${f}_$i.json" 2> "$OUT/weak_w1_${f}_$i.err"
python -c "
import json,sys
l=json.loads([x for x in open('$OUT/weak_w1_${f}_$i.json') if x.startswith('{')][-1])
print('$f', $i, l['ms_per_step'], l['exchange']['slab_bytes'], l['exchange']['format'], l['config'].get('merged_edge_list_consistent'))
"
done
done
python bench.py --kernels-only --steps 20 --warmup 3 > "$OUT/single.json" 2> "$OUT/single.err"
python -c "
import json
l=json.loads([x for x in open('$OUT/single.json') if x.startswith('{')][-1])
print('single', l['ms_per_step'])
"
timeout -k 10 600 python tools/fuzz_gpu_parity.py 80 9000 > "$OUT/fuzz_80_cases_ids3.txt" 2>&1 || { tail -20 "$OUT/fuzz_80_cases_ids3.txt"; exit 1; }
grep -c " ok:" "$OUT/fuzz_80_cases_ids3.txt"
