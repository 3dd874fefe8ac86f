#!/usr/bin/env python3
"""The host graph stage alone on the tables of a job (BASELINE configs[2], or `factor` times it, or the tiled shape), several
times over (MSGPU_GRAPH_DEBUG=1 prints the stage's own phase clock):   python tools/graph_only.py [factor = 1] [tiled]"""
import os
import sys
import time

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from muchsalsa_amd import overlap, synth  # noqa: E402
from muchsalsa_amd.graph import GraphStage  # noqa: E402

factor = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if len(sys.argv) > 2 and sys.argv[2] == "tiled":
    shape = dict(synth.TILED["cfg3"])
    shape["n_reads"] *= factor
else:
    shape = dict(synth.CONFIGS["cfg3"])
    shape["n_reads"] *= factor
    shape["n_anchors"] *= factor
rows, rn, an = synth.accepted_rows(synth.paf_table(**shape))
with overlap.OverlapContext(0) as ctx:
    ctx.set_id_space(len(rn), len(an))
    t, _ = ctx.overlap_batched(rows, 3, resident=True, edgematches=False)
    co = ctx.find_contraction_edges()
    print("%d rows, %d edges, %d contraction edges" % (len(rows), len(t["edges"]), int((co >= 0).sum())), flush=True)
    for rep in range(6 if factor == 1 else 3):
        t0 = time.perf_counter()
        g = GraphStage(t, t["read_len"], t["read_first_line"])
        t1 = time.perf_counter()
        g.clean_up(co, rows)
        t2 = time.perf_counter()
        g.linearize(16)
        t3 = time.perf_counter()
        st = g.stats
        print("create %.1f clean %.1f lin %.1f total %.1f ms; %d components, %d paths, %d path reads" % (
            1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), st.n_components, st.n_paths, st.n_path_reads), flush=True)
        g.close()
