"""Python view of the C-ABI (include/msgpu.h): the call sequence of src/main.cpp:153-178 of the reference.

    paf = parse_paf(path)                      # BlastFileAccessor + BlastFileReader::read  (host)
    ctx = OverlapContext(device=0)             # ThreadPool + Graph + MatchMap construction
    ctx.load_rows(paf.rows)                    # Graph::addVertex + MatchMap::addVertexMatch, in HBM
    ctx.calculate_edges()                      # MatchMap::calculateEdges
    ctx.chaining_and_overlaps()                # the chainingAndOverlaps fan-out
    t = ctx.tables()                           # graph.getEdges() / getEdgeMatches / getEdgeOrders / isShadow

All compute happens in libmsgpu.so on the GPU; this module only moves buffers.  No CPU fallback exists.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import EDGE_DTYPE, EM_DTYPE, ORDER_DTYPE, ROW_DTYPE, Counts, HostTables, Params, Timings


class MsgpuError(RuntimeError):
    def __init__(self, code, detail=""):
        msg = _lib.lib().msgpu_strerror(code).decode()
        super().__init__("%s (%d)%s" % (msg, code, (": " + detail) if detail else ""))
        self.code = code


def default_params():
    p = Params()
    _lib.lib().msgpu_default_params(C.byref(p))
    return p


class Paf:
    """Result of msgpu_parse_paf: accepted rows + the two registries.  `rows` is a view of the loader's own table (no
    copy): page-locked when a GPU is present, so msgpu_load_rows / msgpu_overlap_batched take it at link speed.  The
    view (and every slice of it) keeps the loader's memory alive."""

    def __init__(self, rows, n_lines, read_names, anchor_names, handle=None, n_reads=None, n_anchors=None):
        self.rows = rows
        self.n_lines = n_lines
        self._read_names, self._anchor_names = read_names, anchor_names
        self._handle = handle
        self.n_reads = len(read_names) if n_reads is None else n_reads        # Registry sizes
        self.n_anchors = len(anchor_names) if n_anchors is None else n_anchors

    # the Registry's reverse look-ups as Python lists: built on first use (600 k ctypes calls on BASELINE configs[2] --
    # a quarter second the pipeline never needs: it maps sequence names to ids natively, msgpu_paf_register_sequences)
    @property
    def read_names(self):
        if self._read_names is None:
            L, h = _lib.lib(), self._handle._h
            self._read_names = [L.msgpu_paf_read_name(h, i).decode() for i in range(self.n_reads)]
        return self._read_names

    @property
    def anchor_names(self):
        if self._anchor_names is None:
            L, h = _lib.lib(), self._handle._h
            self._anchor_names = [L.msgpu_paf_anchor_name(h, i).decode() for i in range(self.n_anchors)]
        return self._anchor_names

    def register_sequences(self, kind, seqfile):
        """msgpu_paf_register_sequences: Registry::operator[] for every record of a SeqFile on this PAF's registries
        (kind 0 reads, 1 unitigs) -> (ids per record, id space): what SeqStore.upload takes"""
        ids = np.zeros(len(seqfile), dtype="<u4")
        space = C.c_uint32()
        rc = _lib.lib().msgpu_paf_register_sequences(self._handle._h, int(kind), seqfile._h,
                                                     ids.ctypes.data if len(ids) else None, C.byref(space))
        if rc != 0:
            raise MsgpuError(rc)
        return ids, int(space.value)


class _PafHandle:
    def __init__(self, L, h):
        self._L, self._h = L, h

    def __del__(self):
        if self._h:
            self._L.msgpu_paf_free(self._h)
            self._h = None


def parse_paf(path, params=None):
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.msgpu_parse_paf(os.fsencode(path), C.byref(params) if params is not None else None, C.byref(h))
    if rc != 0:
        raise MsgpuError(rc, str(path))
    owner = _PafHandle(L, h)
    n = C.c_size_t()
    ptr = L.msgpu_paf_rows(h, C.byref(n))
    if n.value:
        buf = (C.c_char * (n.value * ROW_DTYPE.itemsize)).from_address(ptr)
        buf._owner = owner  # numpy keeps `buf` as the base of the view, and with it the loader's table
        rows = np.frombuffer(buf, dtype=ROW_DTYPE, count=n.value)
    else:
        rows = np.zeros(0, dtype=ROW_DTYPE)
    return Paf(rows, L.msgpu_paf_line_count(h), None, None, owner, L.msgpu_paf_read_count(h), L.msgpu_paf_anchor_count(h))


class PinnedRows:
    """A msgpu_row table in page-locked host memory (msgpu_pinned_alloc): what msgpu_overlap_batched wants to be handed
    so that the rows travel to HBM at link speed.  `.array` is a numpy view of it."""

    def __init__(self, rows):
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        self._L = _lib.lib()
        self._p = self._L.msgpu_pinned_alloc(max(rows.nbytes, 1))
        if not self._p:
            raise MemoryError("msgpu_pinned_alloc(%d)" % rows.nbytes)
        buf = (C.c_char * max(rows.nbytes, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=ROW_DTYPE, count=len(rows))
        self.array[:] = rows

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.msgpu_pinned_free(self._p)
            self._p = None

    __del__ = close


class PackedRows:
    """msgpu_pack_rows: a msgpu_row table in the 28-byte form that crosses the host link (page-locked memory; include/msgpu.h
    msgpu_row28 / msgpu_packed_rows): read lengths once per read, lines as runs, flags in the score's top bits.  Raises
    MsgpuError(MSGPU_E_ARG) for a table that does not pack (a score of 2^30 or more, ...)."""

    def __init__(self, rows, n_reads):
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        self._L = _lib.lib()
        self.c = _lib.PackedRows()
        rc = self._L.msgpu_pack_rows(rows.ctypes.data if len(rows) else None, len(rows), int(n_reads), C.byref(self.c))
        if rc != 0:
            self.c = None
            raise MsgpuError(rc)
        self.n_rows, self.n_runs = int(self.c.n_rows), int(self.c.n_runs)
        self.link_bytes = 28 * self.n_rows + 4 * int(n_reads) + 8 * self.n_runs

    def close(self):
        if getattr(self, "c", None) is not None:
            self._L.msgpu_packed_rows_free(C.byref(self.c))
            self.c = None

    __del__ = close


class OverlapContext:
    def __init__(self, device=0, params=None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        rc = self._L.msgpu_create(int(device), C.byref(self.params), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsgpuError(rc)
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msgpu_destroy(self._h)
            self._h = None

    __del__ = close  # (at interpreter shutdown the module globals may be gone already: nothing of them is used above)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MsgpuError(rc, self._L.msgpu_last_error(self._h).decode())

    def set_stream(self, hip_stream_ptr):
        """run on an existing HIP stream (raw hipStream_t value); None / 0 = the context's own stream again"""
        self._check(self._L.msgpu_set_stream(self._h, C.c_void_p(hip_stream_ptr or None)))

    def stream(self):
        """the raw hipStream_t the context queues on (msgpu_get_stream)"""
        return self._L.msgpu_get_stream(self._h) or 0

    def stream_wait(self, hip_stream_ptr):
        """msgpu_stream_wait: the context's stream waits for what is queued on `hip_stream_ptr` so far -- call it before a
        library call that touches device buffers the other stream filled (include/msgpu.h, STREAM CONTRACT rule 3)"""
        self._check(self._L.msgpu_stream_wait(self._h, C.c_void_p(hip_stream_ptr or None)))

    def stream_release(self, hip_stream_ptr):
        """msgpu_stream_release: `hip_stream_ptr` waits for what the context has queued so far -- call it after a library
        call whose device output the other stream is going to read"""
        self._check(self._L.msgpu_stream_release(self._h, C.c_void_p(hip_stream_ptr or None)))

    def set_shard(self, shard, n_shards):
        self._check(self._L.msgpu_set_shard(self._h, shard, n_shards))

    def chain_launches(self):
        """how often chaining_and_overlaps has launched its chain kernels (any thread)"""
        return int(self._L.msgpu_chain_launches(self._h))

    def wait_chain_launch(self, count, timeout_us=1000):
        """block until chain_launches() >= count -> True, or the timeout has passed -> False (any thread; the GIL is released)"""
        h = self._h
        if not h:  # a closed context holds nothing back: the gate is open
            return True
        return self._L.msgpu_wait_chain_launch(h, int(count), int(timeout_us)) == 0

    def set_id_space(self, n_reads, n_anchors):
        """Declare the id counts the loader already knows (Registry sizes); (0, 0) = let the index build find them."""
        self._check(self._L.msgpu_set_id_space(self._h, int(n_reads), int(n_anchors)))

    def load_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        self._check(self._L.msgpu_load_rows(self._h, rows.ctypes.data, len(rows)))

    def load_rows_packed(self, packed):
        """msgpu_load_rows_packed: a PackedRows table (28 bytes per row over the link, expanded in HBM)"""
        self._check(self._L.msgpu_load_rows_packed(self._h, C.byref(packed.c)))

    def load_rows_device(self, dev_ptr, n_rows, keep_alive=None):
        self._keep = keep_alive  # the caller's device buffer must outlive the context's use of it
        self._check(self._L.msgpu_load_rows_device(self._h, C.c_void_p(dev_ptr), n_rows))

    def calculate_edges(self):
        self._check(self._L.msgpu_calculate_edges(self._h))

    def chaining_and_overlaps(self):
        self._check(self._L.msgpu_chaining_and_overlaps(self._h))

    def synchronize(self):
        self._check(self._L.msgpu_synchronize(self._h))

    def counts(self):
        c = Counts()
        self._check(self._L.msgpu_get_counts(self._h, C.byref(c)))
        return c

    def set_stage_events(self, on):
        """mark the stage boundaries with events (timings() per stage) or not (a few microseconds less per marker)"""
        self._check(self._L.msgpu_set_stage_events(self._h, 1 if on else 0))

    def timings(self):
        t = Timings()
        self._check(self._L.msgpu_get_timings(self._h, C.byref(t)))
        return t

    def tables(self):
        """Host copies of the result tables (canonical order, see include/msgpu.h)."""
        c = self.counts()
        edges = np.zeros(c.n_edges, dtype=EDGE_DTYPE)
        ems = np.zeros(c.n_ems, dtype=EM_DTYPE)
        orders = np.zeros(c.n_orders, dtype=ORDER_DTYPE)
        ids = np.zeros(c.n_ids, dtype="<u4")
        self._check(self._L.msgpu_copy_tables(self._h, edges.ctypes.data, ems.ctypes.data, orders.ctypes.data,
                                              ids.ctypes.data))
        return {"edges": edges, "ems": ems, "orders": orders, "ids": ids}

    def overlap_batched(self, rows, n_batches=0, copy=True, resident=False, edgematches=True, device_rows=None):
        """msgpu_overlap_batched[_ex]: rows (numpy table or PinnedRows) -> (tables, info).  The whole overlap path, host
        memory to host memory, as `n_batches` windows of owner reads with the copy of window k behind the compute of
        window k + 1.  tables = the dict of tables(), plus read_len / read_first_line; copy=False returns views of the
        context's pinned result memory (valid until the next call).  resident=True keeps the job's tables whole in HBM
        (find_contraction_edges / get_edgematches / tables() then work on them); edgematches=False (implies resident)
        leaves the EdgeMatch table there: tables["ems"] is None.  device_rows=(device pointer, n_rows): the row table is
        in HBM already (MSGPU_BATCH_ROWS_ON_DEVICE; `rows` is ignored)."""
        h = HostTables()
        flags = (_lib.BATCH_RESIDENT if resident else 0) | (0 if edgematches else _lib.BATCH_NO_EDGEMATCHES)
        if device_rows is not None:
            ptr, n = C.c_void_p(int(device_rows[0]) or None), int(device_rows[1])
            flags |= _lib.BATCH_ROWS_ON_DEVICE
        elif isinstance(rows, PackedRows):  # MSGPU_BATCH_ROWS_PACKED: `rows` = the descriptor
            ptr, n = C.cast(C.byref(rows.c), C.c_void_p), rows.n_rows
            flags |= _lib.BATCH_ROWS_PACKED
        else:
            arr = rows.array if isinstance(rows, PinnedRows) else np.ascontiguousarray(rows, dtype=ROW_DTYPE)
            ptr, n = (arr.ctypes.data if len(arr) else None), len(arr)
        self._check(self._L.msgpu_overlap_batched_ex(self._h, ptr, n, int(n_batches), flags, C.byref(h)))

        def view(ptr, n, dt):
            dt = np.dtype(dt)
            if not n:
                return np.zeros(0, dtype=dt)
            a = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt, count=n)
            return a.copy() if copy else a
        t = {"edges": view(h.edges, h.n_edges, EDGE_DTYPE),
             "ems": view(h.ems, h.n_ems, EM_DTYPE) if edgematches else None,
             "orders": view(h.orders, h.n_orders, ORDER_DTYPE), "ids": view(h.ids, h.n_ids, "<u4"),
             "read_len": view(h.read_len, h.n_reads, "<i4"), "read_first_line": view(h.read_first_line, h.n_reads, "<u4")}
        info = {"n_batches": h.n_batches, "wall_ms": h.wall_ms, "load_ms": h.load_ms,
                "first_batch_ms": h.first_batch_ms, "compute_done_ms": h.compute_done_ms, "n_reads": h.n_reads,
                "n_anchors": h.n_anchors, "n_ems": h.n_ems}
        return t, info

    def get_edgematches(self, edge_idx, copy=True):
        """msgpu_get_edgematches: MatchMap::getEdgeMatches for a list of edge-table indices, out of the EdgeMatch table
        resident in HBM -> (em_off [n + 1], ems)"""
        idx = np.ascontiguousarray(edge_idx, dtype="<u4")
        p_off, p_ems = C.c_void_p(), C.c_void_p()
        self._check(self._L.msgpu_get_edgematches(self._h, idx.ctypes.data if len(idx) else None, len(idx),
                                                  C.byref(p_off), C.byref(p_ems)))
        off = np.frombuffer((C.c_char * ((len(idx) + 1) * 8)).from_address(p_off.value), dtype="<u8", count=len(idx) + 1)
        n = int(off[-1])
        ems = (np.frombuffer((C.c_char * (n * EM_DTYPE.itemsize)).from_address(p_ems.value), dtype=EM_DTYPE, count=n)
               if n else np.zeros(0, dtype=EM_DTYPE))
        return (off.copy(), ems.copy()) if copy else (off, ems)

    def copy_tables_device(self, d_edges=None, d_ems=None, d_orders=None, d_ids=None):
        self._check(self._L.msgpu_copy_tables_device(self._h, C.c_void_p(d_edges), C.c_void_p(d_ems),
                                                     C.c_void_p(d_orders), C.c_void_p(d_ids)))

    def merge_gathered(self, d_gathered, counts, slab_bytes, offs, d_edges, d_orders, d_ids, id_base=None, stream=None):
        """counts: world x 3 (n_edges, n_orders, n_ids); offs: byte offsets (edges, orders, ids) inside a slab; id_base:
        world x 2 (read id base, anchor id base) added to rank r's ids, None = zeros; stream: raw hipStream_t the merge
        kernel runs on, None = the context's (msgpu_merge_gathered_ex)."""
        cnt = np.ascontiguousarray(counts, dtype="<u8")
        base = None if id_base is None else np.ascontiguousarray(id_base, dtype="<u4")
        assert base is None or base.shape == (cnt.shape[0], 2)
        self._check(self._L.msgpu_merge_gathered_ex(self._h, C.c_void_p(d_gathered), cnt.shape[0], cnt.ctypes.data,
                                                    slab_bytes, offs[0], offs[1], offs[2],
                                                    base.ctypes.data if base is not None else None, C.c_void_p(d_edges),
                                                    C.c_void_p(d_orders), C.c_void_p(d_ids), C.c_void_p(stream or None)))

    def pack_wire(self, d_wire_edges, d_wire_orders, d_ids, id_bytes=4):
        """The context's edge / order / id tables in the exchange's wire form (include/msgpu.h) into three device blocks,
        on the context's stream; id_bytes = 3: anchor ids as 24-bit numbers."""
        self._check(self._L.msgpu_pack_wire(self._h, C.c_void_p(d_wire_edges), C.c_void_p(d_wire_orders), C.c_void_p(d_ids),
                                            int(id_bytes)))

    def merge_wire(self, d_gathered, counts, slab_bytes, offs, d_edges, d_orders, d_ids, id_base=None, stream=None, id_bytes=4):
        """merge_gathered over slabs whose blocks are in wire form (msgpu_merge_wire): the same merged tables."""
        cnt = np.ascontiguousarray(counts, dtype="<u8")
        base = None if id_base is None else np.ascontiguousarray(id_base, dtype="<u4")
        assert base is None or base.shape == (cnt.shape[0], 2)
        self._check(self._L.msgpu_merge_wire(self._h, C.c_void_p(d_gathered), cnt.shape[0], cnt.ctypes.data, slab_bytes,
                                             offs[0], offs[1], offs[2], int(id_bytes),
                                             base.ctypes.data if base is not None else None,
                                             C.c_void_p(d_edges), C.c_void_p(d_orders), C.c_void_p(d_ids),
                                             C.c_void_p(stream or None)))

    def find_contraction_edges(self, d_edges=None, n_edges=0, d_orders=None, n_orders=0, n_reads=0):
        """findContractionEdges + sanityCheck (src/main.cpp:416-463, sc.cpp:29-90) on the context's own tables, or on
        device tables given by pointer -> int64 per edge: index of its contraction order, -1 = none."""
        if d_edges is None:
            n_edges = self.counts().n_edges
        out = np.full(int(n_edges), -1, dtype="<i8")
        self._check(self._L.msgpu_find_contraction_edges(self._h, C.c_void_p(d_edges), int(n_edges),
                                                         C.c_void_p(d_orders), int(n_orders), int(n_reads),
                                                         out.ctypes.data if len(out) else None))
        return out

    def reads(self):
        c = self.counts()
        rl = np.zeros(c.n_reads, dtype="<i4")
        fl = np.zeros(c.n_reads, dtype="<u4")
        self._check(self._L.msgpu_copy_reads(self._h, rl.ctypes.data, fl.ctypes.data))
        return rl, fl


class OverlapGroup:
    """msgpu_group: one process, several GPUs behind one call (include/msgpu.h, "a GROUP of contexts").  overlap(rows) ->
    (merged tables, info): every device loads the rows over its own link, builds the index, computes the edges with
    v1 % n == its position; one grouped RCCL all-gather + msgpu_merge_wire merge the edge list on every device."""

    def __init__(self, devices=(0,), params=None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._L.msgpu_group_create(devs, len(devices), C.byref(self.params), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsgpuError(rc)
        self.n = len(devices)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msgpu_group_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_timeout(self, timeout_ms):
        """msgpu_group_set_timeout: overlap() gives up (MsgpuError, code MSGPU_E_TIMEOUT) that long after its entry; 0 = never"""
        rc = self._L.msgpu_group_set_timeout(self._h, int(timeout_ms))
        if rc != 0:
            raise MsgpuError(rc)

    def overlap(self, rows, copy=True):
        arr = rows.array if isinstance(rows, PinnedRows) else np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        t = _lib.GroupTables()
        rc = self._L.msgpu_group_overlap(self._h, arr.ctypes.data if len(arr) else None, len(arr), C.byref(t))
        if rc != 0:
            raise MsgpuError(rc, self._L.msgpu_group_last_error(self._h).decode())

        def view(ptr, n, dt):
            dt = np.dtype(dt)
            if not n:
                return np.zeros(0, dtype=dt)
            a = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt, count=n)
            return a.copy() if copy else a
        tables = {"edges": view(t.edges, t.n_edges, EDGE_DTYPE), "orders": view(t.orders, t.n_orders, ORDER_DTYPE),
                  "ids": view(t.ids, t.n_ids, "<u4"), "read_len": view(t.read_len, t.n_reads, "<i4"),
                  "read_first_line": view(t.read_first_line, t.n_reads, "<u4")}
        info = {k: getattr(t, k) for k in ("n_ems", "n_reads", "n_anchors", "n_members", "id_bytes", "slab_bytes", "wall_ms",
                                           "compute_ms", "exchange_ms", "rows_sliced")}
        return tables, info

    def member_edgematches(self, member, edge_idx_local):
        """MatchMap::getEdgeMatches for edges of member `member`, by their index in THAT member's edge table"""
        ctx = OverlapContext.__new__(OverlapContext)
        ctx._L, ctx._h, ctx._keep = self._L, C.c_void_p(self._L.msgpu_group_ctx(self._h, int(member))), None
        try:
            return ctx.get_edgematches(edge_idx_local)
        finally:
            ctx._h = C.c_void_p()  # (the group owns the context)

    def device_tables(self, member=0):
        """device pointers of the merged tables in member `member`'s HBM: (d_edges, d_orders, d_ids)"""
        p = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
        rc = self._L.msgpu_group_device_tables(self._h, int(member), C.byref(p[0]), C.byref(p[1]), C.byref(p[2]))
        if rc != 0:
            raise MsgpuError(rc)
        return tuple(x.value or 0 for x in p)

    def find_contraction_edges(self, n_edges, n_orders, n_reads, member=0):
        """findContractionEdges + sanityCheck on the merged list in member `member`'s HBM"""
        d_e, d_o, _ = self.device_tables(member)
        out = np.full(int(n_edges), -1, dtype="<i8")
        ctxp = C.c_void_p(self._L.msgpu_group_ctx(self._h, int(member)))
        rc = self._L.msgpu_find_contraction_edges(ctxp, C.c_void_p(d_e), int(n_edges), C.c_void_p(d_o), int(n_orders),
                                                  int(n_reads), out.ctypes.data if len(out) else None)
        if rc != 0:
            raise MsgpuError(rc, self._L.msgpu_last_error(ctxp).decode())
        return out


def build_overlaps(rows, device=0, params=None, shard=0, n_shards=1):
    """rows -> result tables in one call."""
    with OverlapContext(device, params) as ctx:
        if n_shards != 1:
            ctx.set_shard(shard, n_shards)
        ctx.load_rows(rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        return ctx.tables()
