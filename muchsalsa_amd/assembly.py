"""Python view of the assemblePath part of the C-ABI (include/msgpu.h, libms/src/kernel/ap.cpp:615-1362).

    asm = Assembly(store)                       # store: SeqStore holding the nanopore and illumina sequences
    asm.add_path(path, steps, rows, contains, asm_idx)   # host layout of one path -> copy pieces + PAF lines
    asm.finish()                                # ONE gather launch + FASTA wrapping on the device
    asm.text(0 / 1 / 2)                         # temp_1.target.fa / temp_1.query.fa / temp_1.align.paf

`path` = [{"id", "dir", "len"}], `steps` = [{"orders": [{"ids", "score", "base"}], "em": {anchor: (lo, hi)}}] per
consecutive pair of reads (the EdgeOrders / EdgeMatches of diGraph.getEdge(path[i], path[i+1])), `rows` = msgpu_row
table holding the VertexMatches of the path's reads (and of contained reads), `contains` = {read id: [{"nano", "dir",
"anchors": [...]}]} (ContainElements).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (COPY_DTYPE, PATH_CONTAIN_DTYPE, PATH_EM_DTYPE, PATH_INFO_DTYPE, PATH_ORDER_DTYPE, PATH_READ_DTYPE,
                   QUERY_INFO_DTYPE, ROW_DTYPE, PathInput)
from .overlap import MsgpuError

QUERY_KINDS = ("Middle", "Left", "Right", "Contain_Illumina_Match", "Contain_Nano_Middle")


class Assembly:
    def __init__(self, store):
        self._L = _lib.lib()
        self._store = store
        self._h = C.c_void_p()
        rc = self._L.msgpu_assembly_create(store._h, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsgpuError(rc)
        store._assemblies.append(self)  # closed with (before) the store

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msgpu_assembly_free(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise MsgpuError(rc, (self._L.msgpu_assembly_last_error(self._h) or b"").decode())

    def set_rows(self, rows, copy=True):
        """install the VertexMatch table once; later paths may pass rows=None.  copy=False: the library keeps reading the
        caller's array (kept alive here) instead of copying it"""
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        fn = self._L.msgpu_assembly_set_rows if copy else self._L.msgpu_assembly_borrow_rows
        self._check(fn(self._h, rows.ctypes.data if len(rows) else None, len(rows)))
        self._rows_keep = None if copy else rows

    def add_path(self, path, steps, rows, contains=None, asm_idx=1):
        self.add_prepared(self.prepare(path, steps, rows, contains, asm_idx))

    def add_prepared(self, prepared):
        """msgpu_assembly_add_path on an input marshalled by prepare() (so callers can time the layout alone)"""
        self._check(self._L.msgpu_assembly_add_path(self._h, C.byref(prepared[0])))

    def add_prepared_batch(self, prepared, n_threads=1):
        """msgpu_assembly_add_paths: layouts on n_threads host threads, appended in order -> per-path status codes"""
        arr = (PathInput * len(prepared))(*[p[0] for p in prepared])
        status = np.zeros(len(prepared), dtype=np.int32)
        self._check(self._L.msgpu_assembly_add_paths(self._h, arr, len(prepared), int(n_threads), status.ctypes.data))
        return status

    def add_graph_paths(self, graph, n_threads=1):
        """msgpu_assembly_add_graph_paths: every path of a linearised GraphStage, layouts on n_threads host threads ->
        per-path status codes (the graph must stay open until this returns)"""
        status = np.zeros(graph.path_count, dtype=np.int32)
        self._check(self._L.msgpu_assembly_add_graph_paths(self._h, graph._h, int(n_threads),
                                                           status.ctypes.data if len(status) else None))
        return status

    @staticmethod
    def prepare(path, steps, rows, contains=None, asm_idx=1):
        """dict/list description of one path -> (msgpu_path_input, the arrays it points into)"""
        n = len(path)
        reads = np.zeros(n, dtype=PATH_READ_DTYPE)
        for i, p in enumerate(path):
            reads[i] = (p["id"], 2 if p["dir"] is None else (1 if p["dir"] else 0), p["len"])  # e_NONE / e_POS / e_NEG
        order_off, em_off = np.zeros(max(n, 1), dtype="<u4"), np.zeros(max(n, 1), dtype="<u4")
        orders, ids, ems = [], [], []
        for i, st in enumerate(steps):
            for o in st["orders"]:
                orders.append((o["score"], o["base"], len(ids), len(o["ids"]), 0))
                ids.extend(int(x) for x in o["ids"])
            for a, (lo, hi) in st["em"].items():
                ems.append((a, lo, hi))
            order_off[i + 1], em_off[i + 1] = len(orders), len(ems)
        orders = np.array(orders, dtype=PATH_ORDER_DTYPE) if orders else np.zeros(0, dtype=PATH_ORDER_DTYPE)
        ems = np.array(ems, dtype=PATH_EM_DTYPE) if ems else np.zeros(0, dtype=PATH_EM_DTYPE)
        ids = np.asarray(ids, dtype="<u4")
        rows = np.ascontiguousarray(rows if rows is not None else np.zeros(0, dtype=ROW_DTYPE), dtype=ROW_DTYPE)
        cont, canch = [], []
        for host, lst in (contains or {}).items():
            for ce in lst:
                cont.append((host, ce["nano"], 1 if ce["dir"] else 0, len(canch), len(ce["anchors"])))
                canch.extend(int(a) for a in ce["anchors"])
        cont = np.array(cont, dtype=PATH_CONTAIN_DTYPE) if cont else np.zeros(0, dtype=PATH_CONTAIN_DTYPE)
        canch = np.asarray(canch, dtype="<u4")

        def ptr(a):
            return a.ctypes.data if len(a) else None

        inp = PathInput(reads.ctypes.data, n, int(asm_idx), order_off.ctypes.data, ptr(orders), ptr(ids),
                        em_off.ctypes.data, ptr(ems), ptr(rows), len(rows), ptr(cont), len(cont), 0, ptr(canch))
        return inp, (reads, order_off, em_off, orders, ids, ems, rows, cont, canch)

    @property
    def paths(self):
        out = np.zeros(self._L.msgpu_assembly_path_count(self._h), dtype=PATH_INFO_DTYPE)
        for i in range(len(out)):
            self._check(self._L.msgpu_assembly_path_info(self._h, i, out[i:].ctypes.data))
        return out

    @property
    def queries(self):
        out = np.zeros(self._L.msgpu_assembly_query_count(self._h), dtype=QUERY_INFO_DTYPE)
        for i in range(len(out)):
            self._check(self._L.msgpu_assembly_query_info(self._h, i, out[i:].ctypes.data))
        return out

    @property
    def query_count(self):
        return int(self._L.msgpu_assembly_query_count(self._h))

    @property
    def pieces(self):
        n = self._L.msgpu_assembly_pieces(self._h, None, 0)
        out = np.zeros(n, dtype=COPY_DTYPE)
        if n:
            self._L.msgpu_assembly_pieces(self._h, out.ctypes.data, n)
        return out

    @property
    def raw_bytes(self):
        return int(self._L.msgpu_assembly_raw_bytes(self._h))

    def finish(self, stream=None):
        self._check(self._L.msgpu_assembly_finish(self._h, C.c_void_p(stream)))

    def validate(self, band=64):
        """banded edit distance of every query against its PAF window on the contig -> (uint32 per query, DP cells)"""
        out = np.zeros(self._L.msgpu_assembly_query_count(self._h), dtype="<u4")
        cells = C.c_uint64()
        self._check(self._L.msgpu_assembly_validate(self._h, int(band), out.ctypes.data if len(out) else None,
                                                    C.byref(cells)))
        return out, int(cells.value)

    def text(self, which):
        n = C.c_uint64()
        p = self._L.msgpu_assembly_text(self._h, int(which), C.byref(n))
        if not p and which < 2:
            raise MsgpuError(_lib.E_STATE, "finish() first")
        return C.string_at(p, n.value) if n.value else b""

    def text_view(self, which):
        """the same bytes without a copy (valid until the next batch / close): what the pipeline hands to write()"""
        n = C.c_uint64()
        p = self._L.msgpu_assembly_text(self._h, int(which), C.byref(n))
        if not p and which < 2:
            raise MsgpuError(_lib.E_STATE, "finish() first")
        return memoryview((C.c_char * n.value).from_address(p)).cast("B") if n.value else memoryview(b"")
