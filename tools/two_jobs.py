#!/usr/bin/env python3
"""Two independent jobs of the overlap path in flight on one GPU (two contexts, two host threads, two streams): aggregate
overlap-pairs/s against one job at a time.  Shows how much of the device a single job leaves idle.
python tools/two_jobs.py [workload] [steps]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()

    def make():
        ctx = overlap.OverlapContext(device=0)
        ctx.set_id_space(len(rn), len(an))
        ctx.set_stage_events(False)
        return ctx

    def run(ctx, n, out):
        for _ in range(n):
            ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
        ctx.synchronize()
        out.append(ctx.counts().n_edges)

    res = {}
    for jobs in (1, 2, 3):
        ctxs = [make() for _ in range(jobs)]
        for c in ctxs:
            run(c, 3, [])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs, th = [], []
        for c in ctxs:
            t = threading.Thread(target=run, args=(c, steps, outs))
            t.start()
            th.append(t)
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[jobs] = dict(ms_per_job=round(1e3 * dt / (steps * jobs), 4), overlap_pairs_per_s=round(outs[0] * steps * jobs / dt))
        for c in ctxs:
            c.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
