"""msgpu_unpack_wire_host (plain host code of libmsgpu: no GPU): the wire form of a table set back into records on host threads.
Checked against the Python statement of the wire form (muchsalsa_amd.distributed.pack_wire_host / unpack_wire_host) on tables
the oracle produces: whole tables, a window of a larger table set with its bases, 3- and 4-byte ids, ragged id counts, no
records at all."""
import ctypes as C

import numpy as np
import pytest

from muchsalsa_amd import _lib, distributed as D, synth
from muchsalsa_amd._lib import EDGE_DTYPE, ORDER_DTYPE


def _aligned(b, align=8):
    """the bytes of `b` at an `align`-aligned address (the order block holds doubles)"""
    raw = np.zeros(len(b) + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    raw[off: off + len(b)] = b
    return raw[off: off + len(b)], raw


def _unpack(L, blocks, counts, id_bytes, base=None, threads=0, tables=(0,)):
    ne, no, ni = counts
    (eb, keep0), (ob, keep1), (ib, keep2) = (_aligned(np.ascontiguousarray(b)) for b in blocks)
    e, o, ids = np.zeros(ne, dtype=EDGE_DTYPE), np.zeros(no, dtype=ORDER_DTYPE), np.zeros(ni, dtype="<u4")
    e.view(np.uint8)[:] = 0xAB  # every byte must be written, padding included
    o.view(np.uint8)[:] = 0xAB
    b = (C.c_uint64 * 4)(*base) if base is not None else None
    for mask in tables:  # all at once (0), or table by table as a caller does whose blocks arrive one after the other
        rc = L.msgpu_unpack_wire_host(eb.ctypes.data, ob.ctypes.data, ib.ctypes.data if len(ib) else None, id_bytes, ne, no, ni, b,
                                      e.ctypes.data if ne else None, o.ctypes.data if no else None, ids.ctypes.data if ni else None,
                                      threads, mask)
        assert rc == 0
    return {"edges": e, "orders": o, "ids": ids}


@pytest.fixture(scope="module")
def tables(oracle):
    return oracle.overlap(synth.synth_rows(600, 4000, 1500, 5))


@pytest.mark.parametrize("id_bytes", [3, 4])
def test_whole_tables_round_trip(tables, id_bytes):
    L = _lib.lib()
    t = {k: tables[k] for k in ("edges", "orders", "ids")}
    blocks = D.pack_wire_host(t, id_bytes)
    counts = (len(t["edges"]), len(t["orders"]), len(t["ids"]))
    for threads, tables in ((1, (0,)), (0, (0,)), (0, (1, 2, 4)), (3, (4, 1, 2))):
        got = _unpack(L, blocks, counts, id_bytes, threads=threads, tables=tables)
        want = D.unpack_wire_host(*blocks, counts, id_bytes)
        for k in ("edges", "orders", "ids"):
            assert got[k].tobytes() == want[k].tobytes() == t[k].tobytes(), (k, threads, tables)


@pytest.mark.parametrize("id_bytes", [3, 4])
def test_a_window_of_a_larger_table_set_gets_its_bases_back(tables, id_bytes):
    """what the dispatcher does: a window's records point into the JOB's tables; the wire offsets are relative to the window"""
    L = _lib.lib()
    e, o, ids = tables["edges"], tables["orders"], tables["ids"]
    a, b = len(e) // 3, 2 * len(e) // 3 + 1
    o_lo, o_hi = int(e["order_off"][a]), int(e["order_off"][b - 1] + e["order_cnt"][b - 1])
    i_lo, i_hi = int(o["ids_off"][o_lo]), int(o["ids_off"][o_hi - 1] + o["ids_cnt"][o_hi - 1])
    m_lo = int(e["em_off"][a])
    rel = {"edges": e[a:b].copy(), "orders": o[o_lo:o_hi].copy(), "ids": ids[i_lo:i_hi].copy()}
    rel["edges"]["em_off"] -= m_lo
    rel["edges"]["order_off"] -= o_lo
    rel["orders"]["edge_idx"] -= a
    rel["orders"]["ids_off"] -= i_lo
    blocks = D.pack_wire_host(rel, id_bytes)
    got = _unpack(L, blocks, (b - a, o_hi - o_lo, i_hi - i_lo), id_bytes, base=(a, m_lo, o_lo, i_lo))
    assert got["edges"].tobytes() == e[a:b].tobytes()
    assert got["orders"].tobytes() == o[o_lo:o_hi].tobytes()
    assert got["ids"].tobytes() == ids[i_lo:i_hi].tobytes()


@pytest.mark.parametrize("n_ids", [0, 1, 2, 3, 5, 4 * (1 << 15) + 3])
def test_three_byte_ids_of_any_count(n_ids):
    L = _lib.lib()
    rng = np.random.default_rng(n_ids)
    ids = rng.integers(0, 1 << 24, n_ids, dtype=np.uint32)
    t = {"edges": np.zeros(0, dtype=EDGE_DTYPE), "orders": np.zeros(0, dtype=ORDER_DTYPE), "ids": ids}
    eb = np.zeros(8, dtype=np.uint8)
    ob = np.zeros(4, dtype=np.uint8)
    ib = np.zeros((3 * n_ids + 3) // 4 * 4, dtype=np.uint8)
    ib[: 3 * n_ids] = ids.view(np.uint8).reshape(-1, 4)[:, :3].reshape(-1)
    got = _unpack(L, (eb, ob, ib), (0, 0, n_ids), 3)
    assert np.array_equal(got["ids"], ids) and len(got["edges"]) == 0 and len(t["orders"]) == 0


def test_bad_arguments_are_refused():
    L = _lib.lib()
    z = np.zeros(16, dtype=np.uint8)
    assert L.msgpu_unpack_wire_host(z.ctypes.data, z.ctypes.data, None, 5, 0, 0, 0, None, None, None, None, 0, 0) != 0
    assert L.msgpu_unpack_wire_host(None, z.ctypes.data, None, 3, 0, 0, 0, None, None, None, None, 0, 0) != 0
    assert L.msgpu_unpack_wire_host(z.ctypes.data, z.ctypes.data, None, 3, 1, 0, 0, None, None, None, None, 0, 0) != 0
    assert L.msgpu_unpack_wire_host(z.ctypes.data, z.ctypes.data, None, 3, 0, 0, 0, None, None, None, None, 0, 8) != 0
