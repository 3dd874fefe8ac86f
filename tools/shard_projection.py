#!/usr/bin/env python3
"""Per-rank stage times of one cfg3 job when the edges are cut into N shards (run on ONE GPU as shard 0 of N):
what each rank of an N-GPU run spends before the all-gather.  python tools/shard_projection.py [workload]"""
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
        if n > 1:
            ctx.set_shard(0, n)
        best = None
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
        out[n] = dict(ms=best[0], index=best[1], candidates=best[2], chain=best[3], compact=best[4],
                      edges=int(c.n_edges), orders=int(c.n_orders))
        ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
