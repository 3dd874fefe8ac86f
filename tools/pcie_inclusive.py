#!/usr/bin/env python3
"""PCIe-inclusive rate of the overlap path (host row table -> host result tables), for DESIGN.md section 5 (survey_8d_region).
Never bench.py's `value` (that one starts with the rows resident in HBM)."""
import sys
import time

sys.path.insert(0, ".")
from muchsalsa_amd import overlap, synth  # noqa: E402

w = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
rows = synth.synth_rows(**w)
with overlap.OverlapContext(0) as ctx:
    for it in range(3):
        t0 = time.perf_counter()
        ctx.load_rows(rows)
        t1 = time.perf_counter()
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        ctx.synchronize()
        t2 = time.perf_counter()
        t = ctx.tables()
        t3 = time.perf_counter()
        print("iter %d: H2D+index %.2f ms, edges+chain %.2f ms, D2H tables %.2f ms (%.0f MB), total %.2f ms -> %.1f M overlap-pairs/s"
              % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2),
                 sum(v.nbytes for v in t.values()) / 1e6, 1e3 * (t3 - t0), len(t["edges"]) / (t3 - t0) / 1e6))
