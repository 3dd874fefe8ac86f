#!/bin/bash
# kernel trace of the N > 1 step at world 1 (RCCL) with the wire form, and a 2-rank rehearsal (gloo, both ranks on GPU 0) at configs[2] size
set -u
OUT=gpurun_out/r3_04
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 bench.py --force-dist --kernels-only --scaling weak --steps 10 --warmup 2 > "$OUT/weak_w1_under_rocprof.json" 2> "$OUT/weak_w1_under_rocprof.err" || echo "trace failed"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_weak_w1.csv" \;
rm -rf "$OUT/trace"
grep -i "wire\|merge\|pack\|Name" "$OUT/kernel_stats_weak_w1.csv" | cut -c1-200
SECONDS=0
timeout -k 10 500 python3 bench.py --gpus 2 --backend gloo --single-device --kernels-only --steps 5 --warmup 2 > "$OUT/rehearsal_2_ranks_gloo_one_gpu_cfg3_wire.json" 2> "$OUT/rehearsal_2.err" || { echo "rehearsal failed"; tail -20 "$OUT/rehearsal_2.err"; }
echo "rehearsal wall ${SECONDS}s"
python3 - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r3_04/rehearsal_2_ranks_gloo_one_gpu_cfg3_wire.json') if x.startswith('{')][-1])
print(l['n_gpus'], l['scaling'], l['ms_per_step'], l['config'].get('merged_edge_list_consistent'), l['exchange'], l['strong']['merged_edge_list_consistent'], l['strong']['exchange'], l['host_to_host_sharded'].get('ms'))
PY
