"""Python view of the sequence half of the C-ABI (include/msgpu.h): sequence store in HBM + batched gather.

    f = SeqFile(path)                       # SequenceAccessor::buildIndex + whole-record fetch (host)
    store = SeqStore(device=0)
    store.upload(NANOPORE, f, ids)          # records -> HBM, keyed by Registry id
    piece = store.resolve(NANOPORE, read_id, left, right, direction)   # get*Sequence(l, r, d) as a copy piece
    plan = store.plan(pieces)               # batch of pieces with destinations (the layout)
    store.run(plan, d_out_ptr, capacity)    # one kernel launch

Nothing here computes on the CPU except the (l, r) -> (offset, length) arithmetic of strSlice.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import COPY_DTYPE, COPY_ILLUMINA, COPY_REVCOMP  # noqa: F401
from .overlap import MsgpuError

NANOPORE, ILLUMINA = 0, 1


class SeqFile:
    def __init__(self, path, is_fastq=-1, _handle=None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        if _handle is not None:  # (SeqStore.parse_upload: names, lengths and offsets only -- the bytes went to HBM)
            self._h = _handle
            return
        rc = self._L.msgpu_seq_parse(os.fsencode(path), is_fastq, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsgpuError(rc, str(path))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msgpu_seq_free(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __len__(self):
        return self._L.msgpu_seq_count(self._h)

    @property
    def names(self):
        return [self._L.msgpu_seq_name(self._h, i).decode() for i in range(len(self))]

    def sequence(self, i):
        n = self._L.msgpu_seq_length(self._h, i)
        return C.string_at(self._L.msgpu_seq_bases(self._h, i), n) if n else b""

    def length(self, i):
        return int(self._L.msgpu_seq_length(self._h, i))

    def buffer(self):
        """the bytes msgpu_seq_upload sends to HBM: every record's bases plus what lies between them (msgpu_seq_buffer)"""
        n = C.c_uint64()
        p = self._L.msgpu_seq_buffer(self._h, C.byref(n))
        return C.string_at(p, n.value) if n.value and p else b""  # (no bytes on the host after SeqStore.parse_upload)


def str_slice(size, start, end):
    n = C.c_uint64()
    s = _lib.lib().msgpu_str_slice(size, start, end, C.byref(n))
    return int(s), int(n.value)


class ConsensusBase:
    """updateConsensusBase (ap.cpp:205-229) on piece lists: (pieces, borderLeft, borderRight) of a growing contig."""

    def __init__(self):
        self._L = _lib.lib()
        self._h = C.c_void_p(self._L.msgpu_consensus_new())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msgpu_consensus_free(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def update(self, segment, new_lo, new_hi):
        seg = np.ascontiguousarray(segment, dtype=COPY_DTYPE)
        rc = self._L.msgpu_consensus_update(self._h, seg.ctypes.data, len(seg), int(new_lo), int(new_hi))
        if rc != 0:
            raise MsgpuError(rc)

    @property
    def borders(self):
        lo, hi, n = C.c_int32(), C.c_int32(), C.c_uint64()
        self._L.msgpu_consensus_borders(self._h, C.byref(lo), C.byref(hi), C.byref(n))
        return lo.value, hi.value, int(n.value)

    def pieces(self, base=0):
        n = self._L.msgpu_consensus_pieces(self._h, base, None, 0)
        out = np.zeros(n, dtype=COPY_DTYPE)
        if n:
            self._L.msgpu_consensus_pieces(self._h, base, out.ctypes.data, n)
        return out


class SeqStore:
    def __init__(self, device=0):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        rc = self._L.msgpu_seq_create(int(device), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsgpuError(rc)
        self._plans = []
        self._assemblies = []  # muchsalsa_amd.assembly.Assembly objects laid out over this store

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            for a in self._assemblies:
                a.close()
            self._assemblies = []
            for p in self._plans:
                self._L.msgpu_gather_plan_free(p)
            self._plans = []
            self._L.msgpu_seq_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MsgpuError(rc, self._L.msgpu_seq_last_error(self._h).decode())

    def upload(self, kind, seqfile, ids=None, n_ids=0):
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype="<u4")
            self._check(self._L.msgpu_seq_upload(self._h, kind, seqfile._h, ids.ctypes.data, int(n_ids)))
        else:
            self._check(self._L.msgpu_seq_upload(self._h, kind, seqfile._h, None, 0))

    def parse_upload(self, kind, path, is_fastq=-1):
        """msgpu_seq_parse_upload: parse the file and send its bytes to this store in one pass (page-locked ring, no host
        copy of the bases) -> SeqFile without bytes, for Paf.register_sequences + set_ids"""
        h = C.c_void_p()
        self._check(self._L.msgpu_seq_parse_upload(self._h, kind, os.fsencode(path), is_fastq, C.byref(h)))
        return SeqFile(path, _handle=h)

    def upload_bases(self, kind, seqfile):
        """first half of upload: the file's bytes (callable from the thread that parsed the file, one kind per thread)"""
        self._check(self._L.msgpu_seq_upload_bases(self._h, kind, seqfile._h))

    def set_ids(self, kind, seqfile, ids=None, n_ids=0):
        """second half of upload: ids[record] = Registry id of every record of the same file"""
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype="<u4")
            self._check(self._L.msgpu_seq_set_ids(self._h, kind, seqfile._h, ids.ctypes.data, int(n_ids)))
        else:
            self._check(self._L.msgpu_seq_set_ids(self._h, kind, seqfile._h, None, 0))

    def pack_store(self, kind):
        self._check(self._L.msgpu_seq_pack_store(self._h, kind))

    def upload_device(self, kind, d_bases_ptr, n_bases, off, length):
        off = np.ascontiguousarray(off, dtype="<u8")
        length = np.ascontiguousarray(length, dtype="<u8")
        self._check(self._L.msgpu_seq_upload_device(self._h, kind, C.c_void_p(d_bases_ptr), int(n_bases),
                                                    off.ctypes.data, length.ctypes.data, len(off)))

    def pack(self):
        """2 bits per base + exception list for both stores (results of later gathers are unchanged)"""
        self._check(self._L.msgpu_seq_pack(self._h))

    def resolve(self, kind, seq_id, left, right, direction, dst_off=0):
        out = np.zeros(1, dtype=COPY_DTYPE)
        self._check(self._L.msgpu_seq_resolve(self._h, kind, int(seq_id), int(left), int(right), 1 if direction else 0,
                                              out.ctypes.data))
        out["dst_off"] = dst_off
        return out[0]

    # ---- segment builders of assemblePath (ap.cpp:352-579) as piece composers ----------------------------------
    def _row(self, m):
        from ._lib import ROW_DTYPE
        return np.ascontiguousarray(np.asarray(m, dtype=ROW_DTYPE).reshape(1))

    def seg_anchor(self, m, ov, direction):
        r, out, n, ln = self._row(m), np.zeros(1, dtype=COPY_DTYPE), C.c_uint32(), C.c_uint64()
        self._check(self._L.msgpu_seg_anchor(self._h, r.ctypes.data, ov[0], ov[1], 1 if direction else 0,
                                             out.ctypes.data, C.byref(n), C.byref(ln)))
        return out[:n.value], int(ln.value)

    def seg_left_of_anchor(self, m, nanopore_length, ov, direction):
        r, out, n, ln = self._row(m), np.zeros(2, dtype=COPY_DTYPE), C.c_uint32(), C.c_uint64()
        self._check(self._L.msgpu_seg_left_of_anchor(self._h, r.ctypes.data, int(nanopore_length), ov[0], ov[1],
                                                     1 if direction else 0, out.ctypes.data, C.byref(n), C.byref(ln)))
        return out[:n.value], int(ln.value)

    def seg_right_of_anchor(self, m, nanopore_length, ov, direction):
        r, out, n, ln = self._row(m), np.zeros(2, dtype=COPY_DTYPE), C.c_uint32(), C.c_uint64()
        self._check(self._L.msgpu_seg_right_of_anchor(self._h, r.ctypes.data, int(nanopore_length), ov[0], ov[1],
                                                      1 if direction else 0, out.ctypes.data, C.byref(n), C.byref(ln)))
        return out[:n.value], int(ln.value)

    def seg_between_anchors(self, ml, mr, ov_l, ov_r, direction):
        a, b = self._row(ml), self._row(mr)
        out, n, dist, has = np.zeros(3, dtype=COPY_DTYPE), C.c_uint32(), C.c_int32(), C.c_int()
        self._check(self._L.msgpu_seg_between_anchors(self._h, a.ctypes.data, b.ctypes.data, ov_l[0], ov_l[1], ov_r[0],
                                                      ov_r[1], 1 if direction else 0, out.ctypes.data, C.byref(n),
                                                      C.byref(dist), C.byref(has)))
        return out[:n.value], int(dist.value), bool(has.value)

    def plan(self, pieces):
        pieces = np.ascontiguousarray(pieces, dtype=COPY_DTYPE)
        h = C.c_void_p()
        self._check(self._L.msgpu_gather_plan_create(self._h, pieces.ctypes.data, len(pieces), C.byref(h)))
        self._plans.append(h)
        return h

    def plan_out_bytes(self, plan):
        return int(self._L.msgpu_gather_plan_out_bytes(plan))

    def plan_bases(self, plan):
        return int(self._L.msgpu_gather_plan_bases(plan))

    def run(self, plan, d_out_ptr, capacity, stream=None):
        self._check(self._L.msgpu_gather_run(self._h, plan, C.c_void_p(d_out_ptr), int(capacity), C.c_void_p(stream)))

    def synchronize(self):
        self._check(self._L.msgpu_seq_synchronize(self._h))

    def edit_distance(self, d_a_ptr, d_b_ptr, pairs, band):
        """Banded Levenshtein distance of pairs of ranges of two device buffers; band + 1 = 'more than band'."""
        from ._lib import ALIGN_PAIR_DTYPE
        pairs = np.ascontiguousarray(pairs, dtype=ALIGN_PAIR_DTYPE)
        out = np.zeros(len(pairs), dtype="<u4")
        self._check(self._L.msgpu_edit_distance(self._h, C.c_void_p(d_a_ptr), C.c_void_p(d_b_ptr), pairs.ctypes.data,
                                                len(pairs), int(band), out.ctypes.data))
        return out
