#!/usr/bin/env python3
"""Does a memory-bound stage (index, candidates, compaction) run for free beside a vector-issue-bound one (the chain kernels)?
Two contexts on two streams step through the same cfg3 job from two host threads, unsynchronised: the aggregate rate against
one context alone says what a pipeline over windows of one job could gain at best.
  python tools/experiments/two_contexts.py [steps = 20] [n_shards = 1]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_shards = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = WORKLOADS["cfg3"]
rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()


def make(shard):
    ctx = overlap.OverlapContext(device=0)
    ctx.set_id_space(len(rn), len(an))
    st = torch.cuda.Stream(device=0)
    ctx.set_stream(st.cuda_stream)
    if n_shards > 1:
        ctx.set_shard(shard, n_shards)
    return ctx, st


def loop(ctx, st, k, stagger=0.0):
    time.sleep(stagger)
    for _ in range(k):
        ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
    st.synchronize()


a, b = make(0), make(1 % n_shards)
loop(*a, 3)
loop(*b, 3)
t0 = time.perf_counter(); loop(*a, steps); one = (time.perf_counter() - t0) / steps
for stagger in (0.0, 0.0006, 0.0012):
    th = [threading.Thread(target=loop, args=(*a, steps, 0.0)), threading.Thread(target=loop, args=(*b, steps, stagger))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    two = (time.perf_counter() - t0 - stagger) / (2 * steps)
    print("one context %.4f ms/step; two contexts (stagger %.1f ms) %.4f ms/step each => %.3fx" % (1e3 * one, 1e3 * stagger, 1e3 * two, one / two))
