#!/usr/bin/env python3
"""Per-rank stage times of one cfg3 job when the edges are cut into N shards (run on ONE GPU as shard 0 and as the last
shard of N): what each rank of an N-GPU run spends before the all-gather, index build included, and the bytes it would
put into the all-gather.  A projection, not a scaling measurement.  python tools/shard_projection.py [workload]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
    rows, read_names, anchor_names = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
    out = {}
    for n in (1, 2, 4, 8):
        ctx = overlap.OverlapContext(device=0)
        ctx.set_id_space(len(read_names), len(anchor_names))
        best = None
        shard = 0 if len(sys.argv) < 3 else min(int(sys.argv[2]), n - 1)
        if n > 1:
            ctx.set_shard(shard, n)
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            torch.cuda.synchronize()
            dt = 1e3 * (time.perf_counter() - t0)
            if it >= 2 and (best is None or dt < best[0]):
                tm = ctx.timings()
                best = (dt, tm.index_ms, tm.candidates_ms, tm.chain_ms, tm.compact_ms)
        c = ctx.counts()
        out[n] = dict(ms=round(best[0], 4), index=round(best[1], 4), candidates=round(best[2], 4), chain=round(best[3], 4),
                      compact=round(best[4], 4), edges=int(c.n_edges), orders=int(c.n_orders),
                      slab_mb=round((32 * c.n_edges + 64 * c.n_orders + 4 * c.n_ids) / 1e6, 2),
                      gathered_mb=round(n * (32 * c.n_edges + 64 * c.n_orders + 4 * c.n_ids) / 1e6, 1))
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
