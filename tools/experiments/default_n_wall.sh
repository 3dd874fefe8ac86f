#!/bin/bash
# wall time of the default bench line at world 1 over RCCL (--force-dist) and with 2 / 4 real ranks over gloo on one GPU
set -u
OUT=gpurun_out/r4_31
mkdir -p "$OUT"
SECONDS=0
timeout -k 10 500 python3 bench.py --force-dist > "$OUT/default_force_dist.json" 2> "$OUT/default_force_dist.err" || { echo "force-dist failed"; tail -20 "$OUT/default_force_dist.err"; }
echo "force-dist wall ${SECONDS}s"
for n in 2 4; do
SECONDS=0
timeout -k 10 600 python3 bench.py --gpus $n --backend gloo --single-device > "$OUT/default_rehearsal_$n.json" 2> "$OUT/default_rehearsal_$n.err" || { echo "rehearsal $n failed"; tail -20 "$OUT/default_rehearsal_$n.err"; }
echo "rehearsal $n wall ${SECONDS}s"
done
python3 - <<'PY'
import json
for f in ("default_force_dist","default_rehearsal_2","default_rehearsal_4"):
    try:
        l=json.loads([x for x in open('gpurun_out/r4_31/%s.json'%f) if x.startswith('{')][-1])
        print(f, l['n_gpus'], l['scaling'], round(l['ms_per_step'],3), l['config'].get('merged_edge_list_consistent'), l.get('errors'), sorted(k for k in l if isinstance(l[k],dict)))
    except Exception as e: print(f, 'ERR', e)
PY
