#!/usr/bin/env python3
"""One-off differential run on the GPU box: random small workloads (random strands / primary flags, duplicates, shuffled
rows, several coverages) through libmsgpu and through the C oracle; every table must match bit for bit.  Every case also
sends its tables through the exchange's wire form (pack kernel == host statement, wire merge == whole-record merge) and
measures random sequence pairs with msgpu_edit_distance against the full-DP oracle.
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


def wire_roundtrip(ctx, want, rng, what):
    """the exchange's wire form of the context's (resident) tables == the host statement of the form, and the merge of two
    such slabs with id bases == the merge of the whole-record slabs"""
    import torch
    from muchsalsa_amd import distributed as D
    from muchsalsa_amd._lib import EDGE_DTYPE, ORDER_DTYPE
    dev = torch.device("cuda", 0)
    cnt = (len(want["edges"]), len(want["orders"]), len(want["ids"]))
    ib = int(rng.choice([3, 4]))  # 3-byte or 4-byte anchor ids
    nb = D.block_bytes(cnt, wire=ib)
    d = [torch.full((n + 8,), 0xAB, dtype=torch.uint8, device=dev) for n in nb]
    torch.cuda.synchronize()  # include/msgpu.h STREAM CONTRACT rule 3: torch's fills are done before the pack kernel is queued on the context's stream
    ctx.pack_wire(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), id_bytes=ib)
    ctx.synchronize()
    blocks = D.pack_wire_host(want, ib)
    for g, w_, n in zip(d, blocks, nb):
        g = g.cpu().numpy()
        assert g[:n].tobytes() == w_.tobytes() and (g[n:] == 0xAB).all(), what + ": wire block"
    counts = np.array([cnt, cnt], dtype=np.int64)
    id_base = np.array([[0, 0], [int(rng.integers(1, 1 << 20)), int(rng.integers(1, 1 << 20))]], dtype="<u4")
    res = []
    for wire in (False, ib):
        offs, slab_bytes = D.slab_layout(cnt, wire=wire)
        slab = np.full(slab_bytes, 0xCD, dtype=np.uint8)
        for b, off in zip(blocks if wire else [want[k].view(np.uint8) for k in ("edges", "orders", "ids")], offs):
            slab[off: off + len(b)] = b
        d_g = torch.from_numpy(np.concatenate([slab, slab])).to(dev)
        out = [torch.zeros(max(2 * n, 1) * sz, dtype=torch.uint8, device=dev)
               for n, sz in zip(cnt, (EDGE_DTYPE.itemsize, ORDER_DTYPE.itemsize, 4))]
        kw = dict(id_bytes=ib) if wire else {}
        torch.cuda.synchronize()  # (rule 3 again: the zero fills before the merge)
        (ctx.merge_wire if wire else ctx.merge_gathered)(d_g.data_ptr(), counts, slab_bytes, offs, out[0].data_ptr(),
                                                         out[1].data_ptr(), out[2].data_ptr(), id_base=id_base, **kw)
        ctx.synchronize()
        res.append([x.cpu().numpy().tobytes() for x in out])
    assert res[0] == res[1], what + ": wire merge"


def edit_distances(rng, what):
    """msgpu_edit_distance (furthest-reaching kernel) on random pairs against the full-DP oracle, random bound"""
    import torch
    from muchsalsa_amd import sequences as S
    from muchsalsa_amd._lib import ALIGN_PAIR_DTYPE
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    band = int(rng.choice([0, 1, 5, 31, 64, 100, 127]))
    pairs = []
    for _ in range(24):
        a = bytearray(rng.choice(alpha[: int(rng.integers(1, 6))], int(rng.integers(0, 1500))).tobytes())
        b = bytearray(a)
        for _ in range(int(rng.integers(0, band + 8))):
            op, pos = int(rng.integers(0, 3)), int(rng.integers(0, len(b) + 1))
            if op == 0 and b:
                b[min(pos, len(b) - 1)] = int(rng.choice(alpha))
            elif op == 1:
                b.insert(pos, int(rng.choice(alpha)))
            elif b:
                del b[min(pos, len(b) - 1)]
        if rng.random() < 0.15:  # a low-complexity tail: many diagonals slide together
            a += b"A" * int(rng.integers(0, 600))
            b += b"A" * int(rng.integers(0, 600))
        pairs.append((bytes(a), bytes(b)))
    desc = np.zeros(len(pairs), dtype=ALIGN_PAIR_DTYPE)
    ao = bo = 0
    for i, (x, y) in enumerate(pairs):
        desc[i] = (ao, bo, len(x), len(y))
        ao, bo = ao + len(x), bo + len(y)
    da = torch.frombuffer(bytearray(b"".join(p[0] for p in pairs) + b"\0"), dtype=torch.uint8).cuda()
    db = torch.frombuffer(bytearray(b"".join(p[1] for p in pairs) + b"\0"), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with S.SeqStore(0) as st:
        got = st.edit_distance(da.data_ptr(), db.data_ptr(), desc, band)
    assert list(got) == [oracle.edit_distance(x, y, band) for x, y in pairs], what + ": edit distances, bound %d" % band


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    oracle.build()
    group = overlap.OverlapGroup([0])  # msgpu_group_overlap with one member: the C++ multi-GPU path, through RCCL
    # ... and with two and three members sharing this GPU through the rehearsal transport (the switch is read at creation):
    # shards, member threads, sliced row upload + row all-gather, slab layout, pack, merge, slab-wise unpacking on the host
    from muchsalsa_amd import distributed as D
    os.environ["MSGPU_GROUP_TRANSPORT"] = "copy"
    try:
        rehearsal = {k: overlap.OverlapGroup([0] * k) for k in (2, 3)}
    finally:
        del os.environ["MSGPU_GROUP_TRANSPORT"]
    paths = {}
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
        with overlap.OverlapContext(0) as ctx0:
            ctx0.load_rows(feed)
            ctx0.calculate_edges()
            ctx0.chaining_and_overlaps()
            got, ipath = ctx0.tables(), int(ctx0.counts().index_path)
        assert_tables_equal(got, want, what)
        paths[ipath] = paths.get(ipath, 0) + 1
        if ipath == 0:  # the bin path took it: the atomic path must give the same tables (and the per-read Vertex facts)
            os.environ["MSGPU_NO_BIN"] = "1"
            try:
                assert_tables_equal(overlap.build_overlaps(feed), want, what + ", atomic index path")
            finally:
                del os.environ["MSGPU_NO_BIN"]
        if case % 4 == 0:
            gt, ginfo = group.overlap(feed)
            assert_tables_equal(dict(gt, ems=want["ems"]), want, what + ", group of one")
            assert np.array_equal(gt["read_len"], want["read_len"]) and np.array_equal(gt["read_first_line"], want["read_first_line"])
        if case % 4 == 2:
            k = 2 + (case // 4) % 2
            host = D.merge_tables_host([D.shard_view_host(want, r, k) for r in range(k)])
            gt, ginfo = rehearsal[k].overlap(feed)
            assert ginfo["n_members"] == k and ginfo["n_ems"] == len(want["ems"])
            for name in ("edges", "orders", "ids"):
                assert gt[name].tobytes() == host[name].tobytes(), (what, "group of %d" % k, name)
            assert np.array_equal(gt["read_len"], want["read_len"]) and np.array_equal(gt["read_first_line"], want["read_first_line"])
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
            wire_roundtrip(ctx, want, rng, what)
        edit_distances(rng, what)
        scaf = np.bincount(rows["anchor_id"]).max() if len(rows) else 0
        n = want["edges"]["em_cnt"]
        print("case %3d ok: %6d rows %6d edges, EdgeMatches per edge max %3d, orders %6d, longest scaffold %d, index path %d" % (
            case, len(rows), len(n), int(n.max()) if len(n) else 0, len(want["orders"]), int(scaf), ipath), flush=True)
    group.close()
    print("index paths taken (0 bin, 1 atomic, 2 two-pass, +4 generic scaffolds):", dict(sorted(paths.items())), flush=True)


if __name__ == "__main__":
    main()
