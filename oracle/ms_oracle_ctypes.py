"""ctypes binding of the CPU oracle (oracle/libms_oracle.so).

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (muchsalsa_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libms_oracle.so")

ROW_DTYPE = np.dtype([("anchor_id", "<u4"), ("read_id", "<u4"), ("read_len", "<i4"), ("i_lo", "<i4"),
                      ("i_hi", "<i4"), ("n_lo", "<i4"), ("n_hi", "<i4"), ("score", "<u4"), ("line", "<u4"),
                      ("flags", "<u4")])
EDGE_DTYPE = np.dtype([("v1", "<u4"), ("v2", "<u4"), ("em_off", "<u8"), ("order_off", "<u8"), ("em_cnt", "<u4"),
                       ("order_cnt", "<u2"), ("shadow", "u1"), ("pad", "u1")])
EM_DTYPE = np.dtype([("ov_lo", "<i4"), ("ov_hi", "<i4"), ("score", "<f8"), ("anchor_id", "<u4"), ("line", "<u4"),
                     ("flags", "<u4"), ("edge_idx", "<u4")])
ORDER_DTYPE = np.dtype([("edge_idx", "<u4"), ("flags", "<u4"), ("left_offset", "<f8"), ("right_offset", "<f8"),
                        ("score", "<u8"), ("ids_off", "<u8"), ("ids_cnt", "<u4"), ("start", "<u4"), ("end", "<u4"),
                        ("base", "<u4"), ("pad", "<u4", (2,))])
assert ROW_DTYPE.itemsize == 40 and EDGE_DTYPE.itemsize == 32 and EM_DTYPE.itemsize == 32
assert ORDER_DTYPE.itemsize == 64


class Params(C.Structure):
    _fields_ = [("min_matches", C.c_uint32), ("th_length", C.c_uint32), ("th_matches", C.c_uint32),
                ("th_overlap", C.c_uint32), ("wiggle_room", C.c_uint64), ("ratio_pct", C.c_double),
                ("alt_frac", C.c_double)]


class _Rows(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("n_rows", C.c_size_t), ("n_lines", C.c_size_t), ("n_reads", C.c_uint32),
                ("n_anchors", C.c_uint32), ("read_names", C.c_void_p), ("anchor_names", C.c_void_p),
                ("read_names_len", C.c_size_t), ("anchor_names_len", C.c_size_t)]


class _Tables(C.Structure):
    _fields_ = [("edges", C.c_void_p), ("n_edges", C.c_size_t), ("ems", C.c_void_p), ("n_ems", C.c_size_t),
                ("orders", C.c_void_p), ("n_orders", C.c_size_t), ("ids", C.c_void_p), ("n_ids", C.c_size_t),
                ("read_len", C.c_void_p), ("read_first_line", C.c_void_p), ("n_reads", C.c_uint32),
                ("rows_alive", C.c_uint64), ("n_anchors", C.c_uint64), ("p_eval", C.c_uint64),
                ("compat_checks", C.c_uint64), ("shadow_edges", C.c_uint64)]


class _Seqs(C.Structure):
    _fields_ = [("n", C.c_uint32), ("names", C.c_void_p), ("names_len", C.c_size_t), ("bases", C.c_void_p),
                ("off", C.c_void_p)]


_lib = None


def build():
    """Compile oracle/libms_oracle.so with gcc (no-op when up to date)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ms_oracle_default_params.argtypes = [C.POINTER(Params)]
        _lib.ms_oracle_parse_paf.argtypes = [C.c_char_p, C.POINTER(Params), C.POINTER(_Rows)]
        _lib.ms_oracle_parse_paf.restype = C.c_int
        _lib.ms_oracle_free_rows.argtypes = [C.POINTER(_Rows)]
        _lib.ms_oracle_overlap.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Params), C.POINTER(_Tables)]
        _lib.ms_oracle_overlap.restype = C.c_int
        _lib.ms_oracle_free_tables.argtypes = [C.POINTER(_Tables)]
        _lib.ms_oracle_strerror.argtypes = [C.c_int]
        _lib.ms_oracle_strerror.restype = C.c_char_p
        _lib.ms_oracle_seq_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(_Seqs)]
        _lib.ms_oracle_seq_load.restype = C.c_int
        _lib.ms_oracle_seq_free.argtypes = [C.POINTER(_Seqs)]
        _lib.ms_oracle_str_slice.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        _lib.ms_oracle_str_slice.restype = C.c_size_t
        _lib.ms_oracle_get_sequence.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_char_p]
        _lib.ms_oracle_get_sequence.restype = C.c_size_t
        _lib.ms_oracle_anchor_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                                   C.c_char_p]
        _lib.ms_oracle_anchor_sequence.restype = C.c_size_t
        for fn in (_lib.ms_oracle_left_of_anchor, _lib.ms_oracle_right_of_anchor):
            fn.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                           C.c_int, C.c_char_p]
            fn.restype = C.c_size_t
        _lib.ms_oracle_between_anchors.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p,
                                                   C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.POINTER(C.c_int), C.c_char_p,
                                                   C.POINTER(C.c_size_t)]
        _lib.ms_oracle_between_anchors.restype = C.c_int
        _lib.ms_oracle_find_contraction_edges.argtypes = [C.POINTER(_Tables), C.c_uint64, C.c_void_p]
        _lib.ms_oracle_find_contraction_edges.restype = C.c_int
        _lib.ms_oracle_edit_distance.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint32]
        _lib.ms_oracle_edit_distance.restype = C.c_uint32
        _lib.ms_oracle_edit_distance_banded.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint32]
        _lib.ms_oracle_edit_distance_banded.restype = C.c_uint32
    return _lib


def default_params():
    p = Params()
    lib().ms_oracle_default_params(C.byref(p))
    return p


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(lib().ms_oracle_strerror(code).decode())
        self.code = code


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def parse_paf(path, params=None):
    """A1.  Returns dict(rows, n_lines, read_names, anchor_names)."""
    p = params or default_params()
    r = _Rows()
    rc = lib().ms_oracle_parse_paf(os.fsencode(path), C.byref(p), C.byref(r))
    if rc != 0:
        raise OracleError(rc)
    try:
        rows = _copy(r.rows, r.n_rows, ROW_DTYPE)
        rn = C.string_at(r.read_names, r.read_names_len).decode().split("\0")[:-1] if r.read_names_len else []
        an = C.string_at(r.anchor_names, r.anchor_names_len).decode().split("\0")[:-1] if r.anchor_names_len else []
        return {"rows": rows, "n_lines": r.n_lines, "read_names": rn, "anchor_names": an}
    finally:
        lib().ms_oracle_free_rows(C.byref(r))


def overlap(rows, params=None):
    """A1 tail + A2..A7.  Returns dict of canonical tables and counters."""
    p = params or default_params()
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    t = _Tables()
    rc = lib().ms_oracle_overlap(rows.ctypes.data, len(rows), C.byref(p), C.byref(t))
    if rc != 0:
        raise OracleError(rc)
    try:
        return {
            "edges": _copy(t.edges, t.n_edges, EDGE_DTYPE),
            "ems": _copy(t.ems, t.n_ems, EM_DTYPE),
            "orders": _copy(t.orders, t.n_orders, ORDER_DTYPE),
            "ids": _copy(t.ids, t.n_ids, np.dtype("<u4")),
            "read_len": _copy(t.read_len, t.n_reads, np.dtype("<i4")),
            "read_first_line": _copy(t.read_first_line, t.n_reads, np.dtype("<u4")),
            "rows_alive": t.rows_alive, "n_anchors": t.n_anchors, "p_eval": t.p_eval,
            "compat_checks": t.compat_checks, "shadow_edges": t.shadow_edges,
        }
    finally:
        lib().ms_oracle_free_tables(C.byref(t))


def find_contraction_edges(tables, n_reads=None, wiggle=300):
    """findContractionEdges + sanityCheck over result tables (dict with edges, orders) -> int64[n_edges], -1 = none."""
    edges = np.ascontiguousarray(tables["edges"], dtype=EDGE_DTYPE)
    orders = np.ascontiguousarray(tables["orders"], dtype=ORDER_DTYPE)
    if n_reads is None:
        n_reads = int(max(edges["v1"].max(), edges["v2"].max())) + 1 if len(edges) else 0
    t = _Tables()
    t.edges, t.n_edges = edges.ctypes.data, len(edges)
    t.orders, t.n_orders = orders.ctypes.data, len(orders)
    t.n_reads = n_reads
    out = np.full(len(edges), -1, dtype="<i8")
    rc = lib().ms_oracle_find_contraction_edges(C.byref(t), int(wiggle), out.ctypes.data)
    if rc != 0:
        raise OracleError(rc)
    return out


def seq_load(path, is_fastq=-1):
    """SequenceAccessor index + whole-record fetch.  Returns (names, [bytes per record])."""
    sq = _Seqs()
    rc = lib().ms_oracle_seq_load(os.fsencode(path), is_fastq, C.byref(sq))
    if rc != 0:
        raise OracleError(rc)
    try:
        names = C.string_at(sq.names, sq.names_len).decode().split("\0")[:-1] if sq.names_len else []
        off = _copy(sq.off, sq.n + 1, np.dtype("<u8"))
        total = int(off[-1]) if sq.n else 0
        bases = C.string_at(sq.bases, total) if total else b""
        return names, [bases[int(off[i]):int(off[i + 1])] for i in range(sq.n)]
    finally:
        lib().ms_oracle_seq_free(C.byref(sq))


def str_slice(size, start, end):
    """strSlice(original, start, end) -> (offset, length) into the original string."""
    n = C.c_size_t()
    s = lib().ms_oracle_str_slice(size, start, end, C.byref(n))
    return int(s), int(n.value)


def get_sequence(seq, left, right, direction):
    """getNanoporeSequence/getIlluminaSequence(seq, left, right, direction) on a bytes object."""
    out = C.create_string_buffer(len(seq) + 2)
    n = lib().ms_oracle_get_sequence(seq, len(seq), left, right, 1 if direction else 0, out)
    return out.raw[:n]


def _rowp(m):
    r = np.ascontiguousarray(np.asarray(m, dtype=ROW_DTYPE).reshape(1))
    return r, r.ctypes.data


def anchor_sequence(m, illu, ov, direction):
    r, p = _rowp(m)
    out = C.create_string_buffer(len(illu) + 8)
    n = lib().ms_oracle_anchor_sequence(p, illu, len(illu), ov[0], ov[1], 1 if direction else 0, out)
    return out.raw[:n]


def left_of_anchor(m, nano, illu, nanopore_length, ov, direction):
    r, p = _rowp(m)
    out = C.create_string_buffer(len(nano) + len(illu) + 16)
    n = lib().ms_oracle_left_of_anchor(p, nano, len(nano), illu, len(illu), nanopore_length, ov[0], ov[1],
                                       1 if direction else 0, out)
    return out.raw[:n]


def right_of_anchor(m, nano, illu, nanopore_length, ov, direction):
    r, p = _rowp(m)
    out = C.create_string_buffer(len(nano) + len(illu) + 16)
    n = lib().ms_oracle_right_of_anchor(p, nano, len(nano), illu, len(illu), nanopore_length, ov[0], ov[1],
                                        1 if direction else 0, out)
    return out.raw[:n]


def between_anchors(ml, mr, nano, illu_l, illu_r, ov_l, ov_r, direction):
    """-> (distance, sequence or None)"""
    a, pa = _rowp(ml)
    b, pb = _rowp(mr)
    out = C.create_string_buffer(len(nano) + len(illu_l) + len(illu_r) + 24)
    dist, n = C.c_int(), C.c_size_t()
    has = lib().ms_oracle_between_anchors(pa, pb, nano, len(nano), illu_l, len(illu_l), illu_r, len(illu_r), ov_l[0],
                                          ov_l[1], ov_r[0], ov_r[1], 1 if direction else 0, C.byref(dist), out,
                                          C.byref(n))
    return dist.value, (out.raw[:n.value] if has else None)


def edit_distance(a, b, band):
    """min(Levenshtein(a, b), band + 1)"""
    return int(lib().ms_oracle_edit_distance(a, len(a), b, len(b), band))


def edit_distance_banded(a, b, band):
    """Levenshtein distance inside the band |i - j| <= band; band + 1 = 'more than band'.  O(len * band)."""
    return int(lib().ms_oracle_edit_distance_banded(a, len(a), b, len(b), band))
