#!/usr/bin/env python3
"""SURVEY 8(d)'s lean region (rows in pinned host memory -> edge / order / id tables in pinned host memory, EdgeMatch table left
in HBM) at several window counts, with 40-byte and 28-byte rows: median wall time of 7 calls each.
python tools/lean_sweep.py [cfg3]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from muchsalsa_amd import overlap, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rows, rn, an = synth.accepted_rows(synth.paf_table(**synth.CONFIGS[cfg]))
out = {"workload": cfg, "rows": int(len(rows))}
with overlap.OverlapContext(0) as ctx:
    ctx.set_id_space(len(rn), len(an))
    pinned = overlap.PinnedRows(rows)
    packed = overlap.PackedRows(rows, len(rn))
    for name, src in (("rows40", pinned), ("rows28", packed)):
        res = {}
        for b in (2, 3, 4, 5, 6, 8):
            ctx.overlap_batched(src, b, copy=False, resident=True, edgematches=False)
            w, infos = [], []
            for rep in range(7):
                t0 = time.perf_counter()
                _, info = ctx.overlap_batched(src, b, copy=False, resident=True, edgematches=False)
                w.append(1e3 * (time.perf_counter() - t0))
                infos.append(info)
            k = int(np.argsort(w)[len(w) // 2])
            res[b] = {"ms": round(w[k], 3), "load_ms": round(float(infos[k]["load_ms"]), 3), "compute_done_ms": round(float(infos[k]["compute_done_ms"]), 3)}
        out[name] = res
print(json.dumps(out))
