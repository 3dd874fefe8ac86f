#!/bin/bash
# exchange streams at high vs default priority: the N > 1 step at world 1 over RCCL, A/B/A/B on one box
OUT=gpurun_out/r3_04
mkdir -p "$OUT"
for i in 1 2 3; do
for p in 0 2; do
MSGPU_EXCHANGE_PRIORITY=$p python bench.py --force-dist --kernels-only --scaling weak --steps 30 --warmup 3 > "$OUT/prio_${p}_$i.json" 2> "$OUT/prio_${p}_$i.err"
python -c "
import json,sys
l=json.loads([x for x in open('$OUT/prio_${p}_$i.json') if x.startswith('{')][-1])
print('priority', $p, $i, round(l['ms_per_step'],4))
"
done
done
