#!/usr/bin/env python3
"""Steps of ONE shard of an N-shard cfg3 job in a loop (for a kernel trace: what a member of an N-GPU strong-scaling run does
before the exchange).   python tools/shard_step_trace.py [n_shards = 8] [shard = 0] [steps = 8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shard = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w = WORKLOADS["cfg3"]
rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
ctx = overlap.OverlapContext(device=0)
ctx.set_id_space(len(rn), len(an))
if n > 1:
    ctx.set_shard(shard, n)
for _ in range(steps):
    ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
    ctx.calculate_edges()
    ctx.chaining_and_overlaps()
torch.cuda.synchronize()
print(ctx.counts().n_edges)
