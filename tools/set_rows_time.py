#!/usr/bin/env python3
"""msgpu_assembly_set_rows on the cfg3 row table in three input orders (ascending anchor ids = a PAF grouped by query;
grouped by read; shuffled), four fresh assemblies each, milliseconds.  Host only."""
import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from muchsalsa_amd import synth
from muchsalsa_amd.assembly import Assembly
from muchsalsa_amd.sequences import SeqStore
rows,_,_=synth.accepted_rows(synth.paf_table(**synth.CONFIGS["cfg3"]))
store=SeqStore(device=-1)
for name,t in (("anchor order",rows),("by read",rows[np.argsort(rows["read_id"],kind="stable")]),("shuffled",rows[np.random.default_rng(1).permutation(len(rows))])):
    ts=[]
    for i in range(4):
        a=Assembly(store); t0=time.perf_counter(); a.set_rows(t); ts.append(1e3*(time.perf_counter()-t0)); a.close()
    print(name, ["%.1f"%x for x in ts])
