#!/usr/bin/env python3
"""One-off differential run on the GPU box: random small workloads (random strands / primary flags, duplicates, shuffled
rows, several coverages) through libmsgpu and through the C oracle; every table must match bit for bit.
    python tools/fuzz_gpu_parity.py [n_cases] [first_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import ms_oracle_ctypes as oracle  # noqa: E402
from helpers import assert_tables_equal  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    oracle.build()
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        n_reads = int(rng.integers(50, 700))
        read_len = int(rng.integers(2000, 12000))
        cov = int(rng.choice([4, 8, 10, 20, 40]))
        n_anchors = int(rng.integers(max(20, n_reads // 4), n_reads * 6))
        if case % 5 == 4:  # scaffolds far longer than the context pass 1 sorts them in (generic scaffold build)
            cov, n_anchors = int(rng.choice([150, 300])), int(rng.integers(10, 60))
        shape = {}
        if case % 3 == 2:  # the tiled / mixed-length shape: contained orders, contraction edges, the all-compatible shortcut
            shape = dict(tiled=bool(rng.integers(0, 2)), read_len_min=int(read_len // rng.integers(2, 6)))
        rows, _, _ = synth.accepted_rows(synth.paf_table(n_reads, read_len, n_anchors, seed0 + case, coverage=cov, **shape))
        rows = rows.copy()
        mode = case % 4
        if mode >= 1:  # random strands / primary flags
            rows["flags"] = np.where(rng.random(len(rows)) < 0.25, rows["flags"] ^ 1, rows["flags"])
            rows["flags"] = np.where(rng.random(len(rows)) < 0.2, rows["flags"] ^ 2, rows["flags"])
        want = oracle.overlap(rows)
        feed = rows
        if mode >= 2 and len(rows) > 10:  # duplicates with higher line numbers + shuffled rows
            dup = rows[rng.choice(len(rows), min(50, len(rows) // 5), replace=False)].copy()
            dup["line"] = rows["line"].max() + 1 + np.arange(len(dup))
            dup["n_lo"] += 5
            feed = np.concatenate([rows, dup])
            rng.shuffle(feed)
        what = "case %d (reads %d, len %d, anchors %d, cov %d, mode %d)" % (case, n_reads, read_len, n_anchors, cov, mode)
        got = overlap.build_overlaps(feed)
        assert_tables_equal(got, want, what)
        with overlap.OverlapContext(0) as ctx:  # and the same job as windows of owner reads
            nb = int(rng.integers(1, 12))
            got, _ = ctx.overlap_batched(feed, nb)
            assert_tables_equal(got, want, what + ", %d windows" % nb)
            # ... with the job's tables resident and the EdgeMatch table left in HBM (what pipeline.run calls): host tables,
            # the context's own tables, findContractionEdges on them, EdgeMatches of random edges on demand
            nb2 = int(rng.integers(0, 9))
            lean, info = ctx.overlap_batched(feed, nb2, resident=True, edgematches=False)
            assert lean["ems"] is None and info["n_ems"] == len(want["ems"])
            assert_tables_equal(dict(lean, ems=ctx.tables()["ems"]), want, what + ", resident, %d windows" % nb2)
            n_r = len(want["read_len"])
            assert np.array_equal(ctx.find_contraction_edges(), oracle.find_contraction_edges(want, n_r)), what
            if len(want["edges"]):
                idx = rng.integers(0, len(want["edges"]), 40).astype("<u4")
                off, ems = ctx.get_edgematches(idx)
                e = want["edges"][idx]
                exp = np.concatenate([want["ems"][int(o): int(o) + int(c)] for o, c in zip(e["em_off"], e["em_cnt"])])
                assert ems.tobytes() == exp.tobytes() and int(off[-1]) == len(exp), what
        scaf = np.bincount(rows["anchor_id"]).max() if len(rows) else 0
        n = want["edges"]["em_cnt"]
        print("case %3d ok: %6d rows %6d edges, EdgeMatches per edge max %3d, orders %6d, longest scaffold %d" % (
            case, len(rows), len(n), int(n.max()) if len(n) else 0, len(want["orders"]), int(scaf)), flush=True)


if __name__ == "__main__":
    main()
