#!/usr/bin/env python3
"""SURVEY 8(d)'s region alone (rows in pinned host memory -> edge / order / id tables in pinned host memory, EdgeMatch table left
in HBM): best and median wall of msgpu_overlap_batched_ex over a few calls, for A/B runs under different environments.
    python tools/experiments/lean_wall.py [label] [windows = 0 (library's choice)] [all4 = 0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "run"
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
all4 = len(sys.argv) > 3 and sys.argv[3] == "1"
w = WORKLOADS["cfg3"]
rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
DECLARE = os.environ.get("LEAN_DECLARE_IDS", "1") != "0"
pinned = overlap.PinnedRows(rows)
with overlap.OverlapContext(0) as ctx:
    if DECLARE:
        ctx.set_id_space(len(rn), len(an))  # (the parser knows the Registry sizes: what pipeline.run and bench.py do)
    kw = dict(copy=False) if all4 else dict(copy=False, resident=True, edgematches=False)
    ctx.overlap_batched(pinned, windows, **kw)
    walls, infos = [], []
    for _ in range(9):
        t, info = ctx.overlap_batched(pinned, windows, **kw)
        walls.append(info["wall_ms"])
        infos.append(info)
    k = int(np.argmin(walls))
    print("%s: windows %d%s: best %.3f ms, median %.3f (load %.2f, first window %.2f, compute done %.2f) edges %d"
          % (label, infos[k]["n_batches"], " all four tables" if all4 else "", walls[k], float(np.median(walls)), infos[k]["load_ms"],
             infos[k]["first_batch_ms"], infos[k]["compute_done_ms"], len(t["edges"])), flush=True)
pinned.close()
