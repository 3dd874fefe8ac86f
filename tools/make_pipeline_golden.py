#!/usr/bin/env python3
"""Regression fixture of the whole flow: for fixed generated data sets, the SHA-256 of the three output texts produced
by the oracle-only flow (C overlap oracle + C findContractionEdges + Python graph stage + Python assemblePath) and the
stage counts -> tests/golden/pipeline_flow.json.  Run from the repo root: python tools/make_pipeline_golden.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

CASES = [dict(seed=2, jitter=0, n_reads=400, genome_len=250_000), dict(seed=3, jitter=10, n_reads=400, genome_len=250_000),
         dict(seed=11, jitter=6, n_reads=150, genome_len=80_000)]


def flow_digest(case):
    import numpy as np
    import ms_oracle_ctypes as O
    from graphcases import varlen_rows
    from test_gpu_pipeline import oracle_flow
    O.build()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    lay = {}
    rows = varlen_rows(case["n_reads"], 0, case["genome_len"], case["seed"], tiled=True, layout=lay, jitter=case["jitter"])
    genome = np.random.default_rng(99 + case["seed"]).choice(np.frombuffer(b"ACGT", dtype=np.uint8),
                                                             case["genome_len"]).tobytes()
    nano, illu = {}, {}
    for i in range(len(lay["r_start"])):
        s = genome[int(lay["r_start"][i]): int(lay["r_start"][i]) + int(lay["r_len"][i])]
        nano[i] = s if lay["r_fwd"][i] else s.translate(comp)[::-1]
    for j in range(len(lay["a_start"])):
        illu[j] = genome[int(lay["a_start"][j]): int(lay["a_start"][j]) + int(lay["a_len"][j])]
    res = oracle_flow(O, rows, nano, illu)
    texts = [b"".join(r[k] for r in res) for k in ("target_fa", "query_fa", "paf")]
    return dict(case, rows=int(len(rows)), contigs=len(res), target_bases=sum(len(r["target"]) for r in res),
                queries=sum(len(r["queries"]) for r in res),
                sha256=[hashlib.sha256(t).hexdigest() for t in texts])


if __name__ == "__main__":
    out = [flow_digest(c) for c in CASES]
    path = os.path.join(ROOT, "tests", "golden", "pipeline_flow.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, [o["contigs"] for o in out])
