#!/usr/bin/env python3
"""Overlap path at a multiple of cfg3 (GPU box): python tools/scale_overlap.py [factor]
Checks the cross references of the result tables (size-independent properties) and prints the per-step time."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    f = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    t0 = time.perf_counter()
    rows, rn, an = synth.accepted_rows(synth.paf_table(100_000 * f, 10_000, 500_000 * f, 43))
    t_gen = time.perf_counter() - t0
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
    ctx = overlap.OverlapContext(device=0)
    ctx.set_id_space(len(rn), len(an))
    best = None
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it >= 1 and (best is None or dt < best):
            best = dt
    c = ctx.counts()
    t = ctx.tables()
    e, o, ids, em = t["edges"], t["orders"], t["ids"], t["ems"]
    ok = bool(
        np.array_equal(e["order_off"], np.concatenate([[0], np.cumsum(e["order_cnt"])[:-1]]).astype(np.uint64))
        and np.array_equal(e["em_off"], np.concatenate([[0], np.cumsum(e["em_cnt"])[:-1]]).astype(np.uint64))
        and np.array_equal(o["edge_idx"], np.repeat(np.arange(len(e), dtype=np.uint32), e["order_cnt"]))
        and np.array_equal(o["ids_off"], np.concatenate([[0], np.cumsum(o["ids_cnt"])[:-1]]).astype(np.uint64))
        and int(o["ids_cnt"].sum()) == len(ids) and bool(np.all(o["base"] == e["v1"][o["edge_idx"]]))
        and np.array_equal(em["edge_idx"], np.repeat(np.arange(len(e), dtype=np.uint32), e["em_cnt"]))
        and bool(np.all(e["v1"] < e["v2"])) and bool(np.all(np.diff(e["v1"].astype(np.int64)) >= 0)))
    # every order's ids are anchors of its edge's EdgeMatches, in EdgeMatch order (sample of edges)
    rng = np.random.default_rng(1)
    for k in rng.choice(len(e), 2000, replace=False):
        anchors = em["anchor_id"][int(e["em_off"][k]): int(e["em_off"][k]) + int(e["em_cnt"][k])]
        pos = {int(a): i for i, a in enumerate(anchors)}
        for q in range(int(e["order_off"][k]), int(e["order_off"][k]) + int(e["order_cnt"][k])):
            idq = ids[int(o["ids_off"][q]): int(o["ids_off"][q]) + int(o["ids_cnt"][q])]
            p = [pos.get(int(a), -1) for a in idq]
            ok = ok and all(x >= 0 for x in p) and all(p[i] < p[i + 1] for i in range(len(p) - 1))
    print(json.dumps({"factor": f, "rows": int(len(rows)), "reads": int(c.n_reads), "edges": int(c.n_edges),
                      "edgematches": int(c.n_ems), "orders": int(c.n_orders), "ms_per_step": 1e3 * best,
                      "overlap_pairs_per_s": c.n_edges / best, "tables_consistent": ok, "generator_s": round(t_gen, 1)}))


if __name__ == "__main__":
    main()
