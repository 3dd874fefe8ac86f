import sys, time, os, numpy as np
sys.path[:0] = [os.getcwd()]
from muchsalsa_amd import overlap, synth
from muchsalsa_amd.graph import GraphStage
rows, rn, an = synth.accepted_rows(synth.paf_table(**synth.CONFIGS["cfg3"]))
with overlap.OverlapContext(0) as ctx:
    ctx.set_id_space(len(rn), len(an))
    t, _ = ctx.overlap_batched(rows, 3, resident=True, edgematches=False)
    co = ctx.find_contraction_edges()
    for rep in range(6):
        t0=time.perf_counter(); g = GraphStage(t, t["read_len"], t["read_first_line"]); t1=time.perf_counter(); g.clean_up(co, None); t2=time.perf_counter(); g.linearize(16); t3=time.perf_counter()
        print("create %.1f clean %.1f lin %.1f total %.1f ms" % (1e3*(t1-t0),1e3*(t2-t1),1e3*(t3-t2),1e3*(t3-t0)), flush=True)
        g.close()
