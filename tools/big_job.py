#!/usr/bin/env python3
"""A job several times BASELINE configs[2] through the single pass and the resident dispatcher, every table against the C
oracle bit for bit (the oracle takes a minute or two on the box's cores), and the time of a step.
    python tools/big_job.py [factor = 4] [check = 1] [tiled]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from muchsalsa_amd import overlap, synth  # noqa: E402


def wire_round_trip(ctx, host, n_anchors):
    """the resident tables -> the exchange's wire form (msgpu_pack_wire) -> msgpu_merge_wire at world 1 must give the tables
    back byte for byte: at these sizes the blocks and tables pass 2 GiB (the orders of the 32x job: 2.1 GB)"""
    import torch
    from muchsalsa_amd import distributed as D
    from muchsalsa_amd._lib import EDGE_DTYPE, ORDER_DTYPE
    dev = torch.device("cuda", 0)
    cnt = (len(host["edges"]), len(host["orders"]), len(host["ids"]))
    for ib in ((3, 4) if n_anchors <= 1 << 24 else (4,)):
        offs, slab_bytes = D.slab_layout(cnt, wire=ib)
        slab = torch.empty(slab_bytes, dtype=torch.uint8, device=dev)
        t1 = time.perf_counter()
        ctx.pack_wire(slab.data_ptr() + offs[0], slab.data_ptr() + offs[1], slab.data_ptr() + offs[2], id_bytes=ib)
        out = [torch.empty(max(n, 1) * sz, dtype=torch.uint8, device=dev) for n, sz in zip(cnt, (EDGE_DTYPE.itemsize, ORDER_DTYPE.itemsize, 4))]
        ctx.merge_wire(slab.data_ptr(), np.array([cnt], dtype=np.int64), slab_bytes, offs, out[0].data_ptr(), out[1].data_ptr(),
                       out[2].data_ptr(), id_bytes=ib)
        ctx.synchronize()
        dt = time.perf_counter() - t1
        for name, o, n, sz in zip(("edges", "orders", "ids"), out, cnt, (EDGE_DTYPE.itemsize, ORDER_DTYPE.itemsize, 4)):
            want = torch.from_numpy(host[name].view(np.uint8).reshape(-1))
            assert torch.equal(o[: n * sz].cpu(), want), (name, ib)
        whole = sum(n * sz for n, sz in zip(cnt, (EDGE_DTYPE.itemsize, ORDER_DTYPE.itemsize, 4)))
        print("wire form, %d-byte ids: slab %.3f GB against %.3f GB of whole records; pack + merge %.2f ms; the merge gives the "
              "tables back byte for byte" % (ib, slab_bytes / 1e9, whole / 1e9, 1e3 * dt), flush=True)
        del slab, out


def main():
    factor = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    check = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tiled = len(sys.argv) > 3 and sys.argv[3] == "tiled"  # (unitigs that tile the genome, reads of mixed lengths: synth.TILED)
    t0 = time.perf_counter()
    if tiled:
        shape = dict(synth.TILED["cfg3"])
        shape["n_reads"] *= factor
        rows, rn, an = synth.accepted_rows(synth.paf_table(**shape))
    else:
        rows, rn, an = synth.accepted_rows(synth.paf_table(100_000 * factor, 10_000, 500_000 * factor, 43))
    print("%d rows, %d reads, %d anchors (generated in %.1f s)" % (len(rows), len(rn), len(an), time.perf_counter() - t0), flush=True)
    pinned = overlap.PinnedRows(rows)
    with overlap.OverlapContext(0) as ctx:
        ctx.set_id_space(len(rn), len(an))
        ctx.load_rows(pinned.array)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        ctx.synchronize()
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            ctx.load_rows(pinned.array)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            ctx.synchronize()
            ts.append(time.perf_counter() - t1)
        c = ctx.counts()
        print("index path: %d (0 = bin path; it takes one pass per 131,072 reads: %d here)" % (int(c.index_path), (len(rn) + 131071) // 131072), flush=True)
        print("single pass incl. H2D of the rows: %.2f ms; edges %d, EdgeMatches %d, orders %d, ids %d -> %.0f M overlap-pairs/s" % (
            1e3 * min(ts), c.n_edges, c.n_ems, c.n_orders, c.n_ids, c.n_edges / min(ts) / 1e6), flush=True)
        got = ctx.tables() if check else None
        lean, info = ctx.overlap_batched(pinned.array, 0, copy=False, resident=True, edgematches=False)
        t1 = time.perf_counter()
        lean, info = ctx.overlap_batched(pinned.array, 0, copy=False, resident=True, edgematches=False)
        print("resident dispatcher, EdgeMatch table left in HBM, host to host: %.2f ms" % (1e3 * (time.perf_counter() - t1)), flush=True)
        wire_round_trip(ctx, lean, len(an))
        if check:
            import ms_oracle_ctypes as oracle
            from helpers import assert_tables_equal
            oracle.build()
            t1 = time.perf_counter()
            import threading
            done = threading.Event()

            def heartbeat():  # (a long silent run looks hung to whoever watches it)
                while not done.wait(60):
                    print("  ... oracle at work, %.0f s" % (time.perf_counter() - t1), flush=True)
            threading.Thread(target=heartbeat, daemon=True).start()
            want = oracle.overlap(rows)
            done.set()
            print("oracle: %.1f s" % (time.perf_counter() - t1), flush=True)
            assert_tables_equal(got, want, "single pass, factor %d" % factor)
            assert_tables_equal(dict(lean, ems=ctx.tables()["ems"]), want, "resident dispatcher, factor %d" % factor)
            print("every table bit for bit", flush=True)
            co = ctx.find_contraction_edges()  # (on the resident tables)
            assert np.array_equal(co, oracle.find_contraction_edges(want, len(want["read_len"])))
            print("findContractionEdges: %d contraction edges, equal to the oracle's" % int((co >= 0).sum()), flush=True)


if __name__ == "__main__":
    main()
