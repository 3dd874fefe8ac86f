"""msgpu_overlap_batched (the ThreadPool replacement: owner-read batches on two HIP streams, copy of batch k behind the
compute of batch k+1): whatever the batch count, the host tables are the single-pass tables bit for bit."""
import numpy as np
import pytest

from helpers import assert_tables_equal

pytestmark = pytest.mark.gpu


def _single(ctx, rows):
    ctx.load_rows(rows)
    ctx.calculate_edges()
    ctx.chaining_and_overlaps()
    t = ctx.tables()
    t["read_len"], t["read_first_line"] = ctx.reads()
    return t


@pytest.mark.parametrize("shape", [(2000, 5000, 10000, 7), (400, 8000, 3200, 23)])
def test_any_batch_count_gives_the_single_pass_tables(oracle, shape):
    from muchsalsa_amd import overlap, synth
    rows = synth.synth_rows(*shape)
    want = oracle.overlap(rows)
    with overlap.OverlapContext(0) as ctx:
        single = _single(ctx, rows)
        assert_tables_equal(single, want, "single pass")
        pinned = overlap.PinnedRows(rows)
        for b in (1, 2, 3, 8, 37, 0):
            got, info = ctx.overlap_batched(pinned if b != 3 else rows, b)
            assert info["n_batches"] == (b if b else 8)
            assert_tables_equal(got, want, "%d batches" % b)
            assert np.array_equal(got["read_len"], want["read_len"])
            assert np.array_equal(got["read_first_line"], want["read_first_line"])
            assert info["wall_ms"] >= info["compute_done_ms"] >= info["first_batch_ms"] >= info["load_ms"] > 0
        # the context stays usable the ordinary way, and again batched (result memory is reused)
        assert_tables_equal(_single(ctx, rows), want, "single pass after batched runs")
        got, _ = ctx.overlap_batched(pinned, 5, copy=False)
        assert_tables_equal(got, want, "views of the pinned result")
        pinned.close()


def test_batched_edge_cases(oracle):
    """empty input, fewer reads than batches, big edges (> 64 EdgeMatches: side stream), duplicates + shuffled rows
    (generic index build), and a shard's windows."""
    from muchsalsa_amd import distributed as D, overlap, synth
    with overlap.OverlapContext(0) as ctx:
        empty = np.zeros(0, dtype=synth.ROW_DTYPE)
        got, info = ctx.overlap_batched(empty, 4)
        assert all(len(got[k]) == 0 for k in ("edges", "ems", "orders", "ids"))
        tiny = synth.synth_rows(300, 3000, 900, 1)[:5].copy()
        _, tiny["read_id"] = np.unique(tiny["read_id"], return_inverse=True)
        _, tiny["anchor_id"] = np.unique(tiny["anchor_id"], return_inverse=True)
        got, info = ctx.overlap_batched(tiny, 64)
        assert_tables_equal(got, oracle.overlap(tiny), "tiny")
        dense, _, _ = synth.accepted_rows(synth.paf_table(400, 4000, 120, 6, coverage=200))
        want = oracle.overlap(dense)
        assert want["edges"]["em_cnt"].max() > 64
        got, _ = ctx.overlap_batched(dense, 6)
        assert_tables_equal(got, want, "big edges")
        rows = synth.synth_rows(300, 4000, 1000, 9)
        rng = np.random.default_rng(5)
        dup = rows[rng.choice(len(rows), 200, replace=False)].copy()
        dup["line"] = rows["line"].max() + 1 + np.arange(len(dup))
        dup["n_lo"] += 7
        allrows = np.concatenate([rows, dup])
        rng.shuffle(allrows)
        got, _ = ctx.overlap_batched(allrows, 4)
        assert_tables_equal(got, oracle.overlap(rows), "duplicates, shuffled")
        rows = synth.synth_rows(1000, 5000, 4000, 13)
        full = oracle.overlap(rows)
        ctx.set_shard(1, 3)
        got, _ = ctx.overlap_batched(rows, 4)
        assert_tables_equal(got, D.shard_view_host(full, 1, 3), "windows of shard 1/3")
        ctx.set_shard(0, 1)


def test_batched_cfg2_full_size(oracle):
    from muchsalsa_amd import overlap, synth
    rows = synth.synth_rows(**synth.CONFIGS["cfg2"])
    want = oracle.overlap(rows)
    with overlap.OverlapContext(0) as ctx:
        pinned = overlap.PinnedRows(rows)
        got, info = ctx.overlap_batched(pinned, 8)
        assert_tables_equal(got, want, "cfg2, 8 batches")
        got, info2 = ctx.overlap_batched(pinned, 8)  # warm: no allocation left
        assert_tables_equal(got, want, "cfg2, 8 batches, warm")
        print("cfg2 batched: cold %.2f ms, warm %.2f ms (load %.2f, first batch at %.2f, compute done at %.2f)" % (
            info["wall_ms"], info2["wall_ms"], info2["load_ms"], info2["first_batch_ms"], info2["compute_done_ms"]))
        pinned.close()


def _host_edgematches_of(t, idx):
    e = t["edges"][idx]
    off = np.concatenate([[0], np.cumsum(e["em_cnt"].astype(np.uint64))]).astype("<u8")
    parts = [t["ems"][int(o): int(o) + int(c)] for o, c in zip(e["em_off"], e["em_cnt"])]
    return off, (np.concatenate(parts) if parts else np.zeros(0, dtype=t["ems"].dtype))


def test_resident_run_leaves_the_job_tables_in_hbm(oracle):
    """msgpu_overlap_batched_ex(MSGPU_BATCH_RESIDENT): same host tables, and afterwards the context holds the WHOLE job's
    tables like a single pass does -- copy_tables, find_contraction_edges and get_edgematches work on them.
    MSGPU_BATCH_NO_EDGEMATCHES: the EdgeMatch table is not copied; any edge's EdgeMatches come on demand."""
    from muchsalsa_amd import _lib, overlap, synth
    rows = synth.synth_rows(2000, 5000, 10000, 7)
    want = oracle.overlap(rows)
    want_co = oracle.find_contraction_edges(want, len(want["read_len"]))
    rng = np.random.default_rng(3)
    with overlap.OverlapContext(0) as ctx:
        for b in (1, 3, 8, 37):
            got, info = ctx.overlap_batched(rows, b, resident=True)
            assert_tables_equal(got, want, "resident, %d windows: host tables" % b)
            assert_tables_equal(ctx.tables(), want, "resident, %d windows: the context's own tables" % b)
            c = ctx.counts()
            assert (c.n_edges, c.n_ems, c.n_orders, c.n_ids) == tuple(len(want[k]) for k in ("edges", "ems", "orders", "ids"))
            assert c.n_lost_publications == 0  # every size read-back arrived through the mapped-memory publication
            assert np.array_equal(ctx.find_contraction_edges(), want_co)
            with pytest.raises(overlap.MsgpuError) as e:  # no per-edge scratch of the whole job: chaining needs its stage
                ctx.chaining_and_overlaps()
            assert e.value.code == _lib.E_STATE
        got, info = ctx.overlap_batched(overlap.PinnedRows(rows), 5, resident=True, edgematches=False, copy=False)
        assert got["ems"] is None and info["n_ems"] == len(want["ems"])
        for k in ("edges", "orders", "ids"):
            assert got[k].tobytes() == want[k].tobytes(), k
        idx = np.concatenate([[0, len(want["edges"]) - 1], rng.integers(0, len(want["edges"]), 700)]).astype("<u4")
        off, ems = ctx.get_edgematches(idx)
        w_off, w_ems = _host_edgematches_of(want, idx)
        assert np.array_equal(off, w_off) and ems.tobytes() == w_ems.tobytes()
        off, ems = ctx.get_edgematches(np.zeros(0, dtype="<u4"))
        assert list(off) == [0] and len(ems) == 0
        with pytest.raises(overlap.MsgpuError) as e:
            ctx.get_edgematches([len(want["edges"])])
        assert e.value.code == _lib.E_ARG
        assert np.array_equal(ctx.find_contraction_edges(), want_co)
        # the ordinary call sequence afterwards, and the on-demand fetch from ITS resident table
        ctx.load_rows(rows)
        with pytest.raises(overlap.MsgpuError) as e:
            ctx.get_edgematches([0])
        assert e.value.code == _lib.E_STATE
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        assert_tables_equal(ctx.tables(), want, "single pass after resident runs")
        off, ems = ctx.get_edgematches(idx[:50])
        w_off, w_ems = _host_edgematches_of(want, idx[:50])
        assert np.array_equal(off, w_off) and ems.tobytes() == w_ems.tobytes()
        # windows of a shard, resident; big edges (side stream) resident
        full = oracle.overlap(synth.synth_rows(1000, 5000, 4000, 13))
        ctx.set_shard(2, 3)
        got, _ = ctx.overlap_batched(synth.synth_rows(1000, 5000, 4000, 13), 4, resident=True)
        from muchsalsa_amd import distributed as D
        assert_tables_equal(got, D.shard_view_host(full, 2, 3), "resident windows of shard 2/3")
        assert_tables_equal(ctx.tables(), D.shard_view_host(full, 2, 3), "... and the context's tables")
        ctx.set_shard(0, 1)
        dense, _, _ = synth.accepted_rows(synth.paf_table(400, 4000, 120, 6, coverage=200))
        wd = oracle.overlap(dense)
        got, _ = ctx.overlap_batched(dense, 6, resident=True, edgematches=False)
        assert_tables_equal(dict(got, ems=ctx.tables()["ems"]), wd, "big edges, resident")
        big = np.nonzero(wd["edges"]["em_cnt"] > 64)[0].astype("<u4")
        off, ems = ctx.get_edgematches(big)
        w_off, w_ems = _host_edgematches_of(wd, big)
        assert len(big) and np.array_equal(off, w_off) and ems.tobytes() == w_ems.tobytes()


@pytest.mark.parametrize("mode", ["wire", "wire, 4-byte ids", "records"])
def test_lean_windows_leave_in_wire_form_or_as_records(oracle, monkeypatch, mode):
    """MSGPU_BATCH_NO_EDGEMATCHES: the windows' edge / order / id tables cross the host link in the exchange's wire form and a
    host thread turns them back into records (msgpu_unpack_wire_host with the window's bases) -- 3-byte anchor ids while the id
    space fits 24 bits, else the ids as they are; MSGPU_NO_WIRE_COPY=1 sends whole records.  Same host tables, bit for bit, for
    any window count, on a fresh context (host tables grow and move while earlier windows are being unpacked) and a warm one.
    Window cuts follow the visit counts of the index (a grouped input) or the id model (the declared-but-empty anchor ids of
    the 4-byte case make the index generic)."""
    from muchsalsa_amd import overlap, synth
    if mode == "records":
        monkeypatch.setenv("MSGPU_NO_WIRE_COPY", "1")
    rows = synth.synth_rows(2000, 5000, 10000, 7)
    want = oracle.overlap(rows)
    for windows in (1, 2, 3, 7, 0):
        with overlap.OverlapContext(0) as ctx:
            if mode == "wire, 4-byte ids":
                ctx.set_id_space(int(rows["read_id"].max()) + 1, (1 << 24) + 11)
            for rep in ("cold", "warm"):
                got, info = ctx.overlap_batched(overlap.PinnedRows(rows), windows, resident=True, edgematches=False)
                assert got["ems"] is None and info["n_ems"] == len(want["ems"])
                for k in ("edges", "orders", "ids", "read_len", "read_first_line"):
                    assert got[k].tobytes() == want[k].tobytes(), (k, windows, rep)
            assert_tables_equal(ctx.tables(), want, "%s, %d windows: the context's own tables" % (mode, windows))


def test_resident_cfg2_full_size_and_first_call_growth(oracle):
    """cfg2 at full size on a FRESH context: the job tables grow window by window (the kept part of a view moves on
    reallocation) and still come out bit-exact; a warm second call allocates nothing."""
    from muchsalsa_amd import overlap, synth
    rows = synth.synth_rows(**synth.CONFIGS["cfg2"])
    want = oracle.overlap(rows)
    with overlap.OverlapContext(0) as ctx:
        pinned = overlap.PinnedRows(rows)
        got, cold = ctx.overlap_batched(pinned, 8, resident=True, edgematches=False)
        assert_tables_equal(dict(got, ems=ctx.tables()["ems"]), want, "cfg2 resident, cold")
        got, warm = ctx.overlap_batched(pinned, 8, resident=True, edgematches=False, copy=False)
        assert_tables_equal(dict(got, ems=ctx.tables()["ems"]), want, "cfg2 resident, warm")
        print("cfg2 resident without EdgeMatches: cold %.2f ms, warm %.2f ms (load %.2f, compute done at %.2f)" % (
            cold["wall_ms"], warm["wall_ms"], warm["load_ms"], warm["compute_done_ms"]))
        pinned.close()
