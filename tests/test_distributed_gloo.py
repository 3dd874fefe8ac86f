"""N > 1 exchange path on CPU: world_size 2, gloo.  Each rank holds the tables its shard would produce (cut out of
the oracle's single-process tables), runs the real gather protocol of muchsalsa_amd.distributed and merges; the
canonicalised merge must equal the single-process edge list on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import assert_tables_equal


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _three_times(D, t):
    """a rank's tables three times over as ONE dense table set (what a three times larger shard would hold)"""
    m = D.merge_tables_host([t, t, t])
    m["edges"]["em_off"] = np.concatenate([[0], np.cumsum(m["edges"]["em_cnt"].astype(np.uint64))[:-1]])
    return m


def _filler(D, t, wire):
    """fill_slab of the exchange classes for host tables: whole records, or the wire form's three blocks"""
    blocks = D.pack_wire_host(t, wire) if wire else [t[name].view(np.uint8) for name in ("edges", "orders", "ids")]

    def fill(slab, offs):
        for b, off in zip(blocks, offs):
            b = torch.from_numpy(np.ascontiguousarray(b).copy())
            slab[off: off + b.numel()] = b
    return fill


def _worker(rank, world, port, rows_bytes, out_dir, wire=False):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here]
    import ms_oracle_ctypes as oracle
    from muchsalsa_amd import distributed as D
    from muchsalsa_amd._lib import ROW_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = np.frombuffer(rows_bytes, dtype=ROW_DTYPE)
        full = oracle.overlap(rows)
        mine = D.shard_view_host(full, rank, world)

        def fill(slab, offs):
            for name, off in zip(("edges", "orders", "ids"), offs):
                b = torch.from_numpy(mine[name].view(np.uint8).copy())
                slab[off: off + b.numel()] = b

        counts = (len(mine["edges"]), len(mine["orders"]), len(mine["ids"]))
        gathered, all_counts, offs, slab_bytes = D.gather_slabs(counts, fill, torch.device("cpu"))
        per_rank = D.split_gathered_host(gathered.numpy(), all_counts, offs, slab_bytes)
        for ib in (4, 3):  # the wire form loses nothing of a shard's tables, with 4-byte and with 3-byte ids
            for name, blk in zip(("edges", "orders", "ids"), D.unpack_wire_host(*D.pack_wire_host(mine, ib), counts, ib).items()):
                assert blk[1].tobytes() == mine[name].tobytes(), (name, ib)
        merged = D.canonicalize(D.merge_tables_host(per_rank))
        assert int(all_counts[:, 0].sum()) == len(full["edges"])
        # em_off is rank-local by contract; everything else must equal the single-process tables
        want = {k: full[k].copy() for k in ("edges", "orders", "ids")}
        got = {k: merged[k] for k in ("edges", "orders", "ids")}
        want["edges"]["em_off"] = 0
        got["edges"]["em_off"] = 0
        got["ems"] = want["ems"] = np.zeros(0, dtype=full["ems"].dtype)
        assert_tables_equal(got, want, "rank %d" % rank)
        # the one-collective form (header inside the slab, capacity remembered): first call agrees on a capacity, the
        # second goes straight to the slab all-gather, a rank that outgrows the capacity makes every rank repeat
        ex = D.SlabExchange(torch.device("cpu"), wire=wire)
        for call in range(3):
            if call == 2:  # rank 0's tables outgrow the remembered capacity (its own tables three times over)
                grown = _three_times(D, mine) if rank == 0 else mine
            else:
                grown = mine
            cnt = (len(grown["edges"]), len(grown["orders"]), len(grown["ids"]))
            g2, c2, offs2, sb2 = ex.gather(cnt, _filler(D, grown, wire))
            assert np.array_equal(c2[rank], cnt)
            parts = D.split_gathered_host(g2.numpy(), c2, offs2, sb2, wire=wire)
            assert parts[rank]["orders"].tobytes() == grown["orders"].tobytes()
            if call < 2:
                m2 = D.canonicalize(D.merge_tables_host(parts))
                assert m2["orders"].tobytes() == merged["orders"].tobytes() and m2["ids"].tobytes() == merged["ids"].tobytes()
        assert ex.calls == 3 and ex.collectives == 1 + 1 + 1 + 2 and ex.regrows == 1, (ex.calls, ex.collectives, ex.regrows)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _pipelined_worker(rank, world, port, rows_bytes, out_dir, wire=False):
    """PipelinedExchange under gloo: batches submitted one ahead of their collection, a rank that outgrows the capacity in
    two consecutive batches (the second one goes out before the first one's headers were read), id bases in the merge."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here]
    import ms_oracle_ctypes as oracle
    from muchsalsa_amd import distributed as D
    from muchsalsa_amd._lib import ROW_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = np.frombuffer(rows_bytes, dtype=ROW_DTYPE)
        full = oracle.overlap(rows)
        shards = [D.shard_view_host(full, r, world) for r in range(world)]
        big0 = _three_times(D, shards[0])
        # what every rank holds in batch b: the shards; in batches 2 and 3 rank 0's tables are three times as long
        held = lambda b, r: big0 if (r == 0 and b in (2, 3)) else shards[r]  # noqa: E731
        merged = []

        def merge(gathered, allc, offs, slab_bytes, k, stream):
            assert stream is None
            merged.append((D.split_gathered_host(gathered.numpy(), allc, offs, slab_bytes, wire=wire), allc.copy(), slab_bytes))

        pe = D.PipelinedExchange(torch.device("cpu"), merge, wire=wire)
        for b in range(6):
            t = held(b, rank)
            pe.submit((len(t["edges"]), len(t["orders"]), len(t["ids"])), _filler(D, t, wire))
            got = pe.collect()
            assert (got is None) == (b == 0)  # the batch before, from the second submit on
        last = pe.drain()
        assert last is not None and len(merged) == 6
        # The same six batches with the COMMUNICATION THREAD (what a GPU run uses): a grown capacity is adopted at a fixed
        # point -- submit() of batch k + 2 takes what batch k published -- so every batch goes out with the slab size it had
        # without the thread, on every rank, however the two threads interleave (a capacity picked up early on one rank only
        # would be an all-gather of unequal slabs).
        plain, plain_sizes = merged, [m[2] for m in merged]
        for attempt in range(3):
            merged = []
            pt = D.PipelinedExchange(torch.device("cpu"), merge, wire=wire, threaded=True)
            launches, waits = [0], [0]
            if attempt == 2:  # the gate bench.py arms: batch b's all-gather is held until "the chain launch of step b + 1"
                import time

                def gate_wait(token):
                    waits[0] += 1
                    time.sleep(0.001)
                    return launches[0] >= token
                pt.gate_arm, pt.gate_wait = (lambda: launches[0] + 1), gate_wait
            try:
                for b in range(6):
                    launches[0] += 1  # (step b's chain stage starts: the batch before may go)
                    t = held(b, rank)
                    pt.submit((len(t["edges"]), len(t["orders"]), len(t["ids"])), _filler(D, t, wire))
                    assert pt.collect() is None
                    if attempt == 1 and b % 2:
                        import time
                        time.sleep(0.05)  # (another interleaving: the thread gets ahead of the submitter)
                assert pt.drain() is not None and len(merged) == 6  # (drain opens the gate of the last batch)
                assert attempt != 2 or waits[0] >= 1  # (batches still queued when drain() opened the gates skip theirs)
            finally:
                pt.close()
            assert [m[2] for m in merged] == plain_sizes, (attempt, [m[2] for m in merged], plain_sizes)
            assert pt.calls == 6 and pt.regrows == 2 and pt.collectives == 1 + 6 + 2, (pt.calls, pt.regrows, pt.collectives)
            for (parts, allc, _), (parts0, allc0, _) in zip(merged, plain):
                assert np.array_equal(allc, allc0)
                for r in range(world):
                    for k in ("edges", "orders", "ids"):
                        assert parts[r][k].tobytes() == parts0[r][k].tobytes()
        merged = plain
        for b, (parts, allc, _) in enumerate(merged):  # merged in submission order, every rank's tables intact
            for r in range(world):
                want = held(b, r)
                assert tuple(allc[r]) == (len(want["edges"]), len(want["orders"]), len(want["ids"])), (b, r)
                for k in ("edges", "orders", "ids"):
                    assert parts[r][k].tobytes() == want[k].tobytes(), (b, r, k)
        # one agreement + one collective per batch + one repeat each for the two outgrown batches
        assert pe.calls == 6 and pe.regrows == 2 and pe.collectives == 1 + 6 + 2, (pe.calls, pe.regrows, pe.collectives)
        open(os.path.join(out_dir, "pipe%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,wire", [(2, False), (3, False), (2, True), (3, 3)])
def test_pipelined_exchange_one_batch_behind(tmp_path, world, wire):
    from muchsalsa_amd import synth
    rows = synth.synth_rows(300, 4000, 1100, 8)
    port = _free_port()
    mp.spawn(_pipelined_worker, args=(world, port, rows.tobytes(), str(tmp_path), wire), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("pipe%d" % r)) for r in range(world))


@pytest.mark.parametrize("world,wire", [(2, False), (3, False), (2, 3), (3, True)])
def test_gather_and_merge_equals_single_process(tmp_path, world, wire):
    from muchsalsa_amd import synth
    rows = synth.synth_rows(400, 4000, 1500, 5)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, rows.tobytes(), str(tmp_path), wire), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_shard_views_partition_the_edge_list(oracle):
    from muchsalsa_amd import distributed as D, synth
    rows = synth.synth_rows(300, 3000, 900, 1)
    full = oracle.overlap(rows)
    for world in (1, 2, 4, 8):
        shards = [D.shard_view_host(full, r, world) for r in range(world)]
        assert sum(len(s["edges"]) for s in shards) == len(full["edges"])
        assert sum(len(s["ems"]) for s in shards) == len(full["ems"])
        merged = D.canonicalize(D.merge_tables_host(shards))
        assert np.array_equal(merged["edges"]["v1"], full["edges"]["v1"])
        assert np.array_equal(merged["edges"]["v2"], full["edges"]["v2"])
        assert merged["orders"].tobytes() == full["orders"].tobytes() and merged["ids"].tobytes() == full["ids"].tobytes()
