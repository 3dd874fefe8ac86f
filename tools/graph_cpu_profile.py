#!/usr/bin/env python3
"""The host graph stage on the CPU oracle's tables of a workload (no GPU): for working on csrc/graph_stage.cpp in a container
without a device.  Tables are cached under /tmp.   python tools/graph_cpu_profile.py [cfg3|cfg2] [threads] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ms_oracle_ctypes as oracle  # noqa: E402
from muchsalsa_amd import synth  # noqa: E402
from muchsalsa_amd.graph import GraphStage  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cache = "/tmp/graph_tables_%s.npz" % wl
rows, rn, an = synth.accepted_rows(synth.paf_table(**synth.CONFIGS[wl]))
if os.path.exists(cache):
    z = np.load(cache)
    t = {k: z[k] for k in z.files}
else:
    t0 = time.perf_counter()
    t = oracle.overlap(rows)
    t["co"] = oracle.find_contraction_edges(t, len(t["read_len"]))
    np.savez(cache, **{k: v for k, v in t.items() if isinstance(v, np.ndarray)})
    print("oracle: %.1f s" % (time.perf_counter() - t0), flush=True)
co = t["co"]
print("%d rows, %d edges, %d contraction edges" % (len(rows), len(t["edges"]), int((co >= 0).sum())), flush=True)
for rep in range(reps):
    t0 = time.perf_counter()
    g = GraphStage(t, t["read_len"], t["read_first_line"])
    t1 = time.perf_counter()
    g.clean_up(co, rows)
    t2 = time.perf_counter()
    g.linearize(threads)
    t3 = time.perf_counter()
    st = g.stats
    print("create %.1f clean %.1f lin %.1f total %.1f ms; %d components, %d paths, %d path reads" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), st.n_components, st.n_paths, st.n_path_reads), flush=True)
    g.close()
