#!/bin/bash
# kernel timeline of ONE step of shard 0 of 8 of configs[2] (what a member of an 8-GPU strong-scaling run does before the exchange)
OUT=gpurun_out/shard_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > $OUT/run.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from bench import WORKLOADS
from muchsalsa_amd import overlap, synth
w = WORKLOADS["cfg3"]
rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
ctx = overlap.OverlapContext(0)
ctx.set_id_space(len(rn), len(an))
ctx.set_shard(0, 8)
ctx.set_stage_events(False)
for it in range(8):
    ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
    ctx.calculate_edges()
    ctx.chaining_and_overlaps()
ctx.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $OUT/run.py > $OUT/run.log 2>&1 || echo "trace failed"
KT=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$KT" 5 > $OUT/timeline_shard0_of_8.txt
rm -rf "$OUT/trace"
tail -60 $OUT/timeline_shard0_of_8.txt
