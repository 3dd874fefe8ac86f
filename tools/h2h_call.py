#!/usr/bin/env python3
"""A few msgpu_overlap_batched_ex calls on a workload, nothing else (for kernel traces of the dispatcher):
python tools/h2h_call.py [workload] [batches] [flags: full|resident|lean] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "lean"
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rows, rn, an = synth.accepted_rows(synth.paf_table(**synth.CONFIGS[wl]))
pinned = overlap.PinnedRows(rows)
with overlap.OverlapContext(0) as ctx:
    ctx.set_id_space(len(rn), len(an))
    walls = []
    for _ in range(calls):
        t0 = time.perf_counter()
        t, info = ctx.overlap_batched(pinned, B, copy=False, resident=mode != "full", edgematches=mode != "lean")
        walls.append(1e3 * (time.perf_counter() - t0))
    print("%s B=%d %s: walls %s ms; load %.2f first %.2f compute_done %.2f" % (
        wl, B, mode, " ".join("%.2f" % w for w in walls), info["load_ms"], info["first_batch_ms"], info["compute_done_ms"]))
pinned.close()
