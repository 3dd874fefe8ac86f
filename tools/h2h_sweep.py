#!/usr/bin/env python3
"""host_to_host time of the overlap path on a named workload: rows in pinned host memory -> all four tables in pinned
host memory, through msgpu_overlap_batched at several batch counts, next to the single-pass
sequence (msgpu_load_rows + msgpu_calculate_edges + msgpu_chaining_and_overlaps + msgpu_copy_tables to pageable numpy)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    rows, rn, an = synth.accepted_rows(synth.paf_table(**synth.CONFIGS[cfg]))
    out = {"workload": cfg, "rows": int(len(rows))}
    with overlap.OverlapContext(0) as ctx:
        ctx.set_id_space(len(rn), len(an))
        for _ in range(2):
            t0 = time.perf_counter()
            ctx.load_rows(rows)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            t = ctx.tables()
            out["single_pass_pageable_ms"] = 1e3 * (time.perf_counter() - t0)
        out["edges"] = int(len(t["edges"]))
        out["table_bytes"] = int(sum(t[k].nbytes for k in t))
        del t
        pinned = overlap.PinnedRows(rows)
        res = {}
        for b in (1, 2, 4, 8, 16, 32):
            best = None
            for rep in range(4):
                _, info = ctx.overlap_batched(pinned, b, copy=False)
                if rep and (best is None or info["wall_ms"] < best["wall_ms"]):
                    best = info
            res[b] = {k: round(float(v), 3) for k, v in best.items() if k.endswith("_ms")}
        out["batched"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
