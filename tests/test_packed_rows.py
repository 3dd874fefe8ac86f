"""The row table's 28-byte link form (include/msgpu.h: msgpu_row28 / msgpu_packed_rows, reference row content
BlastFileReader.cpp:101-126).  Host side here (msgpu_pack_rows: plain host code, no GPU): the packed block holds what a reader
needs to rebuild every 40-byte row -- checked by rebuilding them in numpy -- and refuses what does not pack.  The GPU side
(msgpu_load_rows_packed, the expansion kernel, MSGPU_BATCH_ROWS_PACKED) is in the -m gpu tests below."""
import ctypes as C

import numpy as np
import pytest

from muchsalsa_amd import _lib, overlap, synth
from muchsalsa_amd._lib import ROW_DTYPE


def _unpack_host(p):
    """numpy statement of k_expand_rows"""
    c = p.c
    n, V, R = int(c.n_rows), int(c.n_reads), int(c.n_runs)
    raw = np.frombuffer((C.c_char * (28 * n)).from_address(c.rows), dtype="<u4").reshape(n, 7) if n else np.zeros((0, 7), "<u4")
    read_len = np.frombuffer((C.c_char * (4 * V)).from_address(c.read_len), dtype="<i4") if V else np.zeros(0, "<i4")
    rs = np.frombuffer((C.c_char * (4 * R)).from_address(c.run_start), dtype="<u4")
    rd = np.frombuffer((C.c_char * (4 * R)).from_address(c.run_delta), dtype="<u4")
    out = np.zeros(n, dtype=ROW_DTYPE)
    out["anchor_id"], out["read_id"] = raw[:, 0], raw[:, 1]
    out["read_len"] = read_len[raw[:, 1]]
    for k, f in enumerate(("i_lo", "i_hi", "n_lo", "n_hi")):
        out[f] = raw[:, 2 + k].view("<i4")
    out["score"] = raw[:, 6] & 0x3fffffff
    out["flags"] = raw[:, 6] >> 30
    idx = np.arange(n, dtype=np.uint32)
    out["line"] = idx + rd[np.searchsorted(rs, idx, side="right") - 1]
    return out


def test_pack_rows_round_trip_and_refusals():
    tab = synth.paf_table(600, 4000, 1500, 5)
    rows, read_names, _ = synth.accepted_rows(tab)
    assert len(rows) > 5000 and int(rows["line"].max()) >= len(rows)  # (rejected lines in between: several runs)
    p = overlap.PackedRows(rows, len(read_names))
    assert p.n_runs > 1 and int(p.c.n_rows) == len(rows)
    back = _unpack_host(p)
    # read_len: the packed form carries the FIRST line's per read (Graph.cpp:148); the generator gives every line of a read the same
    assert back.tobytes() == rows.tobytes()
    assert p.link_bytes < 0.72 * rows.nbytes
    p.close()
    # a table whose lines are the row indices: one run
    plain = rows.copy()
    plain["line"] = np.arange(len(plain))
    q = overlap.PackedRows(plain, len(read_names))
    assert q.n_runs == 1 and _unpack_host(q).tobytes() == plain.tobytes()
    q.close()
    # empty table
    e = overlap.PackedRows(rows[:0], 0)
    assert int(e.c.n_rows) == 0
    e.close()
    # what does not pack: a score that needs the flag bits, a read id beyond the per-read table, a line below its row index
    for field, value in (("score", 1 << 30), ("read_id", len(read_names)), ("flags", 4)):
        bad = rows.copy()
        bad[field][len(bad) // 2] = value
        with pytest.raises(overlap.MsgpuError) as info:
            overlap.PackedRows(bad, len(read_names))
        assert info.value.code == _lib.E_ARG
    bad = rows.copy()
    bad["line"][100] = 3
    with pytest.raises(overlap.MsgpuError):
        overlap.PackedRows(bad, len(read_names))


@pytest.mark.gpu
def test_load_rows_packed_gives_the_same_tables(oracle):
    """msgpu_load_rows_packed (28 bytes per row over the link, k_expand_rows in HBM) == msgpu_load_rows on the same table, every
    field of every table and the per-read Vertex facts; through the dispatcher too (MSGPU_BATCH_ROWS_PACKED, EdgeMatches left in
    HBM: what pipeline.run calls)."""
    from helpers import assert_tables_equal
    for shape in ((1000, 5000, 4000, 13), (37, 3000, 150, 4), (3000, 2500, 9000, 5)):
        rows, read_names, anchor_names = synth.accepted_rows(synth.paf_table(*shape))
        want = oracle.overlap(rows)
        p = overlap.PackedRows(rows, len(read_names))
        with overlap.OverlapContext(0) as ctx:
            ctx.set_id_space(len(read_names), len(anchor_names))
            ctx.load_rows_packed(p)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            got, reads = ctx.tables(), ctx.reads()
            assert_tables_equal(got, want, "packed rows %r" % (shape,))
            ctx.load_rows(rows)
            assert all(np.array_equal(a, b) for a, b in zip(reads, ctx.reads()))
            lean, _ = ctx.overlap_batched(p, 3, resident=True, edgematches=False)
            assert lean["ems"] is None
            assert_tables_equal(dict(lean, ems=ctx.tables()["ems"]), want, "packed rows through the dispatcher %r" % (shape,))
            full, _ = ctx.overlap_batched(p, 4)
            assert_tables_equal(full, want, "packed rows, four windows, all tables %r" % (shape,))
        p.close()
