"""Python view of the host graph stage of the C-ABI (include/msgpu.h): graph clean-up, connected components,
getDirectedGraph and linearizeGraph -- everything between the overlap tables and assemblePath.

    g = GraphStage(tables, read_len, read_first_line)     # tables of OverlapContext.tables(), reads of .reads()
    g.clean_up(contraction_order, rows)                   # contraction_order from OverlapContext.find_contraction_edges
    g.linearize()
    g.set_path_edgematches(*ctx.get_edgematches(g.path_edges()))   # only when tables["ems"] is None (EdgeMatches in HBM)
    for i in range(g.path_count): asm.add_prepared((g.path_input(i), g))
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EDGE_DTYPE, EM_DTYPE, ORDER_DTYPE, ROW_DTYPE, GraphStats, PathInput
from .overlap import MsgpuError


class GraphStage:
    def __init__(self, tables, read_len, read_first_line):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        e = np.ascontiguousarray(tables["edges"], dtype=EDGE_DTYPE)
        # ems None: the EdgeMatch table stayed in HBM; the path edges' EdgeMatches follow through set_path_edgematches
        m = None if tables.get("ems") is None else np.ascontiguousarray(tables["ems"], dtype=EM_DTYPE)
        o = np.ascontiguousarray(tables["orders"], dtype=ORDER_DTYPE)
        i = np.ascontiguousarray(tables["ids"], dtype="<u4")
        rl = np.ascontiguousarray(read_len, dtype="<i4")
        fl = np.ascontiguousarray(read_first_line, dtype="<u4")
        self.n_edges, self.n_reads = len(e), len(rl)
        self._tables = (e, m, o, i)  # msgpu_graph_create_borrowed: the tables live as long as this object

        def ptr(a):
            return a.ctypes.data if a is not None and len(a) else None
        if m is not None and len(m) == 0 and len(e):  # (a NULL EdgeMatch table means "on demand": keep an empty one apart)
            m = np.zeros(1, dtype=EM_DTYPE)
            self._tables = (e, m, o, i)
        rc = self._L.msgpu_graph_create_borrowed(ptr(e), len(e), ptr(m), 0 if m is None else len(m), ptr(o), len(o),
                                                 ptr(i), len(i), ptr(rl), ptr(fl), len(rl), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsgpuError(rc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msgpu_graph_free(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise MsgpuError(rc, (self._L.msgpu_graph_last_error(self._h) or b"").decode())

    def clean_up(self, contraction_order, rows=None):
        co = np.ascontiguousarray(contraction_order, dtype="<i8")
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        self._check(self._L.msgpu_graph_clean_up(self._h, co.ctypes.data if len(co) else None,
                                                 rows.ctypes.data if rows is not None and len(rows) else None,
                                                 0 if rows is None else len(rows)))

    def linearize(self, threads=1):
        self._check(self._L.msgpu_graph_set_threads(self._h, int(threads)))
        self._check(self._L.msgpu_graph_linearize(self._h))

    @property
    def stats(self):
        s = GraphStats()
        self._check(self._L.msgpu_graph_get_stats(self._h, C.byref(s)))
        return s

    @property
    def path_count(self):
        return int(self._L.msgpu_graph_path_count(self._h))

    def path_edges(self):
        """edge-table index of the edge under every path step, paths concatenated (msgpu_graph_path_edges)"""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.msgpu_graph_path_edges(self._h, C.byref(p), C.byref(n)))
        if not n.value:
            return np.zeros(0, dtype="<u4")
        return np.frombuffer(C.string_at(p.value, n.value * 4), dtype="<u4").copy()

    def set_path_edgematches(self, em_off, ems):
        off = np.ascontiguousarray(em_off, dtype="<u8")
        m = np.ascontiguousarray(ems, dtype=EM_DTYPE)
        self._check(self._L.msgpu_graph_set_path_edgematches(self._h, off.ctypes.data, m.ctypes.data if len(m) else None))

    def path_input(self, i):
        p = PathInput()
        self._check(self._L.msgpu_graph_path_input(self._h, int(i), C.byref(p)))
        return p

    def state(self):
        va, vd = np.zeros(self.n_reads, dtype=np.uint8), np.zeros(self.n_reads, dtype=np.uint8)
        ea, ec = np.zeros(self.n_edges, dtype=np.uint8), np.zeros(self.n_edges, dtype=np.uint8)
        ew = np.zeros(self.n_edges, dtype="<u8")
        self._check(self._L.msgpu_graph_state(self._h, va.ctypes.data, vd.ctypes.data, ea.ctypes.data, ec.ctypes.data,
                                              ew.ctypes.data))
        return dict(vertex_alive=va.astype(bool), vertex_direction=vd, edge_alive=ea.astype(bool), edge_consensus=ec,
                    edge_weight=ew)

    def path(self, i):
        """path i decoded into the dict form of muchsalsa_amd.assembly (tests / inspection)"""
        from ._lib import PATH_CONTAIN_DTYPE, PATH_EM_DTYPE, PATH_ORDER_DTYPE, PATH_READ_DTYPE
        p = self.path_input(i)

        def view(ptr, n, dt):
            if not n:
                return np.zeros(0, dtype=dt)
            return np.frombuffer(C.string_at(ptr, n * np.dtype(dt).itemsize), dtype=dt)
        reads = view(p.reads, p.n_reads, PATH_READ_DTYPE)
        ooff, eoff = view(p.order_off, p.n_reads, "<u4"), view(p.em_off, p.n_reads, "<u4")
        orders, ems = view(p.orders, int(ooff[-1]), PATH_ORDER_DTYPE), view(p.ems, int(eoff[-1]), PATH_EM_DTYPE)
        path = [{"id": int(r["read_id"]), "dir": (False, True, None)[int(r["direction"])],
                 "len": int(r["nanopore_length"])} for r in reads]
        steps = []
        for s in range(p.n_reads - 1):
            os_ = []
            for o in orders[ooff[s]:ooff[s + 1]]:
                ids = view(p.ids + 4 * int(o["ids_off"]), int(o["ids_cnt"]), "<u4")
                os_.append({"ids": [int(x) for x in ids], "score": int(o["score"]), "base": int(o["base_read"])})
            steps.append({"orders": os_, "em": {int(m["anchor_id"]): (int(m["ov_lo"]), int(m["ov_hi"]))
                                                  for m in ems[eoff[s]:eoff[s + 1]]}})
        cont = view(p.contains, p.n_contains, PATH_CONTAIN_DTYPE)
        n_anch = int(max((int(c["anchors_off"]) + int(c["anchors_cnt"]) for c in cont), default=0))
        anch = view(p.contain_anchors, n_anch, "<u4")
        contains = {}
        for c in cont:
            contains.setdefault(int(c["host_read"]), []).append(
                dict(nano=int(c["nano"]), dir=bool(c["direction"]),
                     anchors=[int(a) for a in anch[int(c["anchors_off"]): int(c["anchors_off"]) + int(c["anchors_cnt"])]]))
        return path, steps, contains
