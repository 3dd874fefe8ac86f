"""Multi-GPU exchange of the overlap tables: one process per GPU, torch.distributed (RCCL over xGMI; gloo on CPU).

Sharding: every rank loads the full row table and owns the edges whose first vertex v1
satisfies v1 % world == rank (msgpu_set_shard).  There is no collective on the data path before the results exist;
the only exchange is ONE all-gather of the per-rank (edges | orders | ids) slab ("merge the edge list").

`gather_slabs` is the protocol (device-agnostic: CUDA tensors + nccl on GPUs, CPU tensors + gloo in tests).
On GPUs the gathered slabs are compacted and re-based by the HIP kernel behind msgpu_merge_gathered;
`merge_tables_host` is the host-side statement of the same merge used to check it.
"""
import numpy as np

from ._lib import EDGE_DTYPE, ORDER_DTYPE

ALIGN = 256  # sub-table alignment inside a slab (bytes)


def shard_of(v1, world):
    """Owner rank of an edge: its first vertex modulo the world size."""
    return v1 % world


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


def wire_id_bytes(wire):
    """wire: False / 0 = whole records; True / 4 = wire form with 4-byte ids; 3 = wire form with 3-byte ids -> 0, 4 or 3"""
    return 0 if not wire else (3 if int(wire) == 3 else 4)


def block_bytes(counts, wire=False):
    """Bytes of a rank's (edge, order, id) blocks: whole records, or the wire form of include/msgpu.h (17 n + 8, 33 n + 4,
    4 n or -- wire = 3 -- 3 n rounded up to a word)."""
    ne, no, ni = (int(x) for x in counts)
    ib = wire_id_bytes(wire)
    if ib:
        return 17 * ne + 8, 33 * no + 4, (4 * ni if ib == 4 else (3 * ni + 3) // 4 * 4)
    return ne * EDGE_DTYPE.itemsize, no * ORDER_DTYPE.itemsize, ni * 4


def slab_layout(max_counts, wire=False):
    """Byte offsets of (edges, orders, ids) inside a slab sized for the largest rank, and the slab size."""
    be, bo, bi = block_bytes(max_counts, wire)
    off_e = 0
    off_o = _round_up(off_e + be)
    off_i = _round_up(off_o + bo)
    size = _round_up(off_i + bi)
    return (off_e, off_o, off_i), max(size, ALIGN)


def pack_wire_host(t, id_bytes=4):
    """Host statement of msgpu_pack_wire: a rank's {edges, orders, ids} -> (edge block, order block, id block) as uint8
    arrays.  Asserts what the wire form relies on: dense tables, start / end / base given by flags and edge, zero padding."""
    e, o, ids = t["edges"], t["orders"], np.ascontiguousarray(t["ids"], dtype="<u4")
    ne, no = len(e), len(o)

    def csr(off, cnt, total=None):
        off, cnt = off.astype(np.uint64), cnt.astype(np.uint64)
        dense = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
        first = off[0] if len(off) else np.uint64(0)  # (a block of a larger table may start anywhere; it must be dense)
        assert np.array_equal(off, dense[:-1] + first), "the wire form needs dense tables"
        assert int(first + dense[-1]) < 2 ** 32
        assert total is None or not len(off) or int(first) == 0 and int(dense[-1]) == total
        return (dense + first).astype("<u4")
    v1, v2 = e["v1"][o["edge_idx"]], e["v2"][o["edge_idx"]]
    sv1 = (o["flags"] & 1).astype(bool)
    assert np.array_equal(o["start"], np.where(sv1, v1, v2)) and np.array_equal(o["end"], np.where(sv1, v2, v1))
    assert np.array_equal(o["base"], v1) and not o["pad"].any() and not e["pad"].any() and int(o["flags"].max(initial=0)) < 256
    eb = np.concatenate([csr(e["em_off"], e["em_cnt"]).view(np.uint8), csr(e["order_off"], e["order_cnt"], no).view(np.uint8),
                         e["v1"].astype("<u4").view(np.uint8), e["v2"].astype("<u4").view(np.uint8), e["shadow"].astype(np.uint8)])
    ob = np.concatenate([o["left_offset"].astype("<f8").view(np.uint8), o["right_offset"].astype("<f8").view(np.uint8),
                         o["score"].astype("<u8").view(np.uint8), csr(o["ids_off"], o["ids_cnt"], len(ids)).view(np.uint8),
                         o["edge_idx"].astype("<u4").view(np.uint8), o["flags"].astype(np.uint8)])
    assert len(eb) == 17 * ne + 8 and len(ob) == 33 * no + 4
    if wire_id_bytes(id_bytes) == 3:  # the low three bytes of every id, rounded up to a word with zeros
        assert int(ids.max(initial=0)) < 1 << 24
        ib = np.zeros((3 * len(ids) + 3) // 4 * 4, dtype=np.uint8)
        ib[: 3 * len(ids)] = ids.view(np.uint8).reshape(-1, 4)[:, :3].reshape(-1)
        return eb, ob, ib
    return eb, ob, ids.view(np.uint8)


def unpack_wire_host(eb, ob, ib, counts, id_bytes=4):
    """Host statement of what msgpu_merge_wire reconstructs for ONE rank (no re-basing): blocks -> {edges, orders, ids}."""
    ne, no, ni = (int(x) for x in counts)
    eb, ob = np.ascontiguousarray(eb[: 17 * ne + 8]), np.ascontiguousarray(ob[: 33 * no + 4])
    u4 = lambda b, lo, n: b[lo: lo + 4 * n].view("<u4")  # noqa: E731
    em_off, order_off = u4(eb, 0, ne + 1), u4(eb, 4 * (ne + 1), ne + 1)
    e = np.zeros(ne, dtype=EDGE_DTYPE)
    e["v1"], e["v2"] = u4(eb, 8 * (ne + 1), ne), u4(eb, 8 * (ne + 1) + 4 * ne, ne)
    e["em_off"], e["em_cnt"] = em_off[:-1], np.diff(em_off.astype(np.int64))
    e["order_off"], e["order_cnt"] = order_off[:-1], np.diff(order_off.astype(np.int64))
    e["shadow"] = eb[8 * (ne + 1) + 8 * ne: 8 * (ne + 1) + 9 * ne]
    o = np.zeros(no, dtype=ORDER_DTYPE)
    o["left_offset"], o["right_offset"] = ob[: 8 * no].view("<f8"), ob[8 * no: 16 * no].view("<f8")
    o["score"] = ob[16 * no: 24 * no].view("<u8")
    ids_off = u4(ob, 24 * no, no + 1)
    o["ids_off"], o["ids_cnt"] = ids_off[:-1], np.diff(ids_off.astype(np.int64))
    o["edge_idx"] = u4(ob, 24 * no + 4 * (no + 1), no)
    o["flags"] = ob[24 * no + 4 * (no + 1) + 4 * no: 24 * no + 4 * (no + 1) + 5 * no]
    v1, v2 = e["v1"][o["edge_idx"]], e["v2"][o["edge_idx"]]
    sv1 = (o["flags"] & 1).astype(bool)
    o["start"], o["end"], o["base"] = np.where(sv1, v1, v2), np.where(sv1, v2, v1), v1
    if wire_id_bytes(id_bytes) == 3:
        ids = np.zeros((ni, 4), dtype=np.uint8)
        ids[:, :3] = np.ascontiguousarray(ib[: 3 * ni]).reshape(ni, 3)
        return {"edges": e, "orders": o, "ids": ids.reshape(-1).view("<u4").copy()}
    return {"edges": e, "orders": o, "ids": np.ascontiguousarray(ib[: 4 * ni]).view("<u4").copy()}


def gather_slabs(counts, fill_slab, device, group=None):
    """The exchange step.

    counts     -- (n_edges, n_orders, n_ids) of this rank
    fill_slab  -- callable(slab_uint8_tensor, offs) that writes this rank's three tables into the slab
    returns (gathered uint8 tensor [world * slab_bytes], all_counts int64 ndarray [world, 3], offs, slab_bytes)
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=device)
    allc = torch.empty(world * 3, dtype=torch.int64, device=device)  # flat: gloo rejects a 2-D output
    dist.all_gather_into_tensor(allc, mine, group=group)
    all_counts = allc.cpu().numpy().reshape(world, 3)
    offs, slab_bytes = slab_layout(all_counts.max(axis=0))
    slab = torch.empty(slab_bytes, dtype=torch.uint8, device=device)  # padding between the tables is never read
    fill_slab(slab, offs)
    gathered = torch.empty(world * slab_bytes, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(gathered, slab, group=group)  # the one collective of the path
    return gathered, all_counts, offs, slab_bytes


HEADER = 256  # bytes in front of a slab: int64 {n_edges, n_orders, n_ids} of the sending rank


class SlabExchange:
    """The exchange step as ONE collective per call (north_star: "a single RCCL all-gather over xGMI to merge the edge
    list").  gather_slabs() needs two -- the ranks first tell each other their table sizes so that everyone can size the
    slab.  Here every slab starts with a header carrying its sender's counts, and the slab capacity is remembered from
    call to call: the first call agrees on it with one small all-gather, later calls go straight to the slab
    all-gather.  A rank whose tables outgrew the capacity sends its header alone; every rank reads the same headers, so
    all of them enlarge the capacity and repeat the collective together (counted in `regrows`).  Identical decisions on
    every rank by construction: nothing but gathered data enters them."""

    def __init__(self, device, group=None, slack=1.03125, wire=False):
        self.device, self.group, self.slack, self.wire = device, group, slack, wire
        self.cap = None
        self.calls = self.collectives = self.regrows = 0
        self.slab_bytes = 0

    def _layout(self):
        offs, size = slab_layout(self.cap, self.wire)
        return tuple(HEADER + o for o in offs), HEADER + size

    def _agree(self, counts):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        mine = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=self.device)
        allc = torch.empty(world * 3, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        self.collectives += 1
        self._grow(allc.cpu().numpy().reshape(world, 3))

    def _grow(self, all_counts):
        need = all_counts.max(axis=0)
        self.cap = tuple(int(n * self.slack) + 64 for n in need)

    def gather(self, counts, fill_slab):
        """counts = (n_edges, n_orders, n_ids) of this rank; fill_slab(slab, offs) writes the three tables at the byte
        offsets offs.  -> (gathered, all_counts [world, 3], offs, slab_bytes): the arguments msgpu_merge_gathered takes."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        self.calls += 1
        if self.cap is None:
            self._agree(counts)
        while True:
            offs, slab_bytes = self._layout()
            self.slab_bytes = slab_bytes
            # the slab, the gathered block and the page-locked header are kept from call to call (a step of the strong-scaling
            # bench is under a millisecond per member at N = 8: an allocation, a pageable copy and a tensor per call showed in it)
            if getattr(self, "_bufs", None) is None or self._bufs[0].numel() != slab_bytes or self._bufs[1].numel() != world * slab_bytes:
                self._bufs = (torch.empty(slab_bytes, dtype=torch.uint8, device=self.device),  # padding is never read
                              torch.empty(world * slab_bytes, dtype=torch.uint8, device=self.device),
                              torch.empty(3, dtype=torch.int64, pin_memory=(self.device.type == "cuda")))
            slab, gathered, head = self._bufs
            head[0], head[1], head[2] = int(counts[0]), int(counts[1]), int(counts[2])
            slab[:24].view(torch.int64).copy_(head, non_blocking=True)
            if all(int(c) <= k for c, k in zip(counts, self.cap)):
                fill_slab(slab, offs)
            dist.all_gather_into_tensor(gathered, slab, group=self.group)  # the one collective of the path
            self.collectives += 1
            heads = gathered.view(world, slab_bytes)[:, :24].contiguous().view(torch.int64).cpu().numpy().reshape(world, 3)
            if (heads <= np.asarray(self.cap, dtype=np.int64)[None, :]).all():
                return gathered, heads.astype(np.int64), offs, slab_bytes
            self._grow(heads)
            self.regrows += 1


def split_gathered_host(gathered, all_counts, offs, slab_bytes, wire=False):
    """Host view of a gathered buffer: list of per-rank {edges, orders, ids} numpy tables."""
    buf = np.asarray(gathered, dtype=np.uint8)
    out = []
    for r, (ne, no, ni) in enumerate(np.asarray(all_counts, dtype=np.int64)):
        base = r * slab_bytes
        if wire:
            out.append(unpack_wire_host(buf[base + offs[0]:], buf[base + offs[1]:], buf[base + offs[2]:], (ne, no, ni), wire))
            continue
        out.append({
            "edges": buf[base + offs[0]: base + offs[0] + ne * EDGE_DTYPE.itemsize].view(EDGE_DTYPE).copy(),
            "orders": buf[base + offs[1]: base + offs[1] + no * ORDER_DTYPE.itemsize].view(ORDER_DTYPE).copy(),
            "ids": buf[base + offs[2]: base + offs[2] + ni * 4].view("<u4").copy(),
        })
    return out


def merge_tables_host(per_rank):
    """Rank-major merge with re-based cross references (what msgpu_merge_gathered produces on the GPU)."""
    eb = ob = ib = 0
    edges, orders, ids = [], [], []
    for t in per_rank:
        e, o = t["edges"].copy(), t["orders"].copy()
        e["order_off"] += ob
        o["edge_idx"] += eb
        o["ids_off"] += ib
        edges.append(e)
        orders.append(o)
        ids.append(t["ids"])
        eb += len(e)
        ob += len(o)
        ib += len(t["ids"])
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)  # noqa: E731
    return {"edges": cat(edges, EDGE_DTYPE), "orders": cat(orders, ORDER_DTYPE), "ids": cat(ids, "<u4")}


def canonicalize(merged):
    """Sort a merged (rank-major) edge list into the canonical single-GPU order: edges by (v1, v2), orders and ids
    following their edges.  em_off is rank-local and left untouched (callers comparing with a single-GPU table
    should ignore it).  Used by tests to prove that any GPU count gives the same edge list."""
    e, o, ids = merged["edges"], merged["orders"], merged["ids"]
    perm = np.lexsort((e["v2"], e["v1"]))
    e2 = e[perm].copy()
    o_parts, id_parts = [], []
    ob = ib = 0
    for k, src in enumerate(perm):
        n = int(e[src]["order_cnt"])
        oo = o[int(e[src]["order_off"]): int(e[src]["order_off"]) + n].copy()
        e2[k]["order_off"] = ob
        for q in range(n):
            c = int(oo[q]["ids_cnt"])
            id_parts.append(ids[int(oo[q]["ids_off"]): int(oo[q]["ids_off"]) + c])
            oo[q]["ids_off"] = ib
            oo[q]["edge_idx"] = k
            ib += c
        o_parts.append(oo)
        ob += n
    return {"edges": e2,
            "orders": np.concatenate(o_parts) if o_parts else np.zeros(0, dtype=ORDER_DTYPE),
            "ids": np.concatenate(id_parts) if id_parts else np.zeros(0, dtype="<u4")}


def shard_view_host(tables, shard, world):
    """The tables a shard produces, cut out of single-GPU tables (edges with v1 % world == shard, dense, re-based).
    Test helper: states what msgpu_set_shard(shard, world) must return."""
    e, em, o, ids = tables["edges"], tables["ems"], tables["orders"], tables["ids"]
    keep = np.nonzero(e["v1"] % world == shard)[0]
    e2 = e[keep].copy()
    em_parts, o_parts, id_parts = [], [], []
    mb = ob = ib = 0
    for k, src in enumerate(keep):
        ne, no = int(e[src]["em_cnt"]), int(e[src]["order_cnt"])
        m = em[int(e[src]["em_off"]): int(e[src]["em_off"]) + ne].copy()
        m["edge_idx"] = k
        em_parts.append(m)
        oo = o[int(e[src]["order_off"]): int(e[src]["order_off"]) + no].copy()
        for q in range(no):
            c = int(oo[q]["ids_cnt"])
            id_parts.append(ids[int(oo[q]["ids_off"]): int(oo[q]["ids_off"]) + c])
            oo[q]["ids_off"] = ib
            oo[q]["edge_idx"] = k
            ib += c
        o_parts.append(oo)
        e2[k]["em_off"], e2[k]["order_off"] = mb, ob
        mb += ne
        ob += no
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)  # noqa: E731
    return {"edges": e2, "ems": cat(em_parts, em.dtype), "orders": cat(o_parts, ORDER_DTYPE), "ids": cat(id_parts, "<u4")}


class PipelinedExchange:
    """The one-collective exchange of SlabExchange, taken OFF the compute stream and off the compute thread: a batch's slab
    is filled on the caller's stream, and its all-gather, header check and merge run on a communication stream of their
    own, issued by a communication thread -- so batch k's merge over xGMI overlaps batch k + 1's candidate scan and
    chaining on the GPU, and the host calls of the exchange overlap the host's waits inside libmsgpu (the reference's
    ThreadPool keeps its workers busy the same way while a finished phase's results are consumed,
    libms/src/threading/ThreadPool.cpp:38-129).

        pe = PipelinedExchange(device, merge)       # merge(gathered, all_counts, offs, slab_bytes, slot, stream)
        for every batch:
            ... compute ...
            pe.submit(counts, fill_slab)            # slab k % 2 filled on the caller's stream; the rest happens behind
            pe.collect()                            # (without a communication thread: finishes the batch BEFORE)
        pe.drain()                                  # everything submitted is merged

    Capacity protocol = SlabExchange's (header in the slab, capacity remembered, identical decisions on every rank because
    only gathered data enters them -- and because a grown capacity is ADOPTED AT A FIXED POINT: the communication side keeps
    its own capacity (`_comm_cap`, grown while it checks headers, in batch order) and publishes it into the finished slot;
    submit() of batch k + 2 adopts what batch k published, after it has waited for batch k -- never what batch k + 1 may or
    may not have published by then.  So batch k + 1 always goes out with the capacity batch k was sent with, with or without
    the communication thread, on every rank).  The capacity is the largest rank's tables + 1/32: every rank's slab is padded to it
    and the padding travels, so slack is xGMI time in every step, while outgrowing it costs one repeated collective once.  A rank that outgrows the capacity sends its header alone and keeps its tables in a
    private stash; when the headers are read every rank enlarges the capacity, re-lays its own slab of THAT batch (still
    intact: a slot is reused two batches later; or the stash) and repeats the collective.  Collectives are issued in
    submission order by one thread, so every rank issues the same sequence.
    On CPU tensors (gloo, the tests) there are no streams and no thread: the all-gather completes inside submit() and
    collect() finishes the batch before the one just submitted, which keeps the one-batch-behind bookkeeping honest."""

    def __init__(self, device, merge, group=None, slack=1.03125, threaded=None, wire=False):
        import torch
        self.device, self.group, self.slack, self.merge, self.wire = device, group, slack, merge, wire
        self.cuda = device.type == "cuda"
        # default priority: MSGPU_EXCHANGE_PRIORITY=1 puts this stream (the exchange's small copies and its merge) ahead of
        # the next batch's kernels -- measured at world 1 that costs the step 0.045 ms (the merge then takes bandwidth from
        # the compute instead of filling its gaps; profiles/r3_04/priority_ab.txt); RCCL's own stream is high-priority
        # by default (bench.py, mode 2; 0 = neither)
        import os
        prio = -1 if os.environ.get("MSGPU_EXCHANGE_PRIORITY", "2") == "1" else 0
        self.comm = torch.cuda.Stream(device=device, priority=prio) if self.cuda else None
        # the communication thread: default on a GPU; on CPU tensors (gloo) only on request -- the tests' rehearsal of the
        # threaded capacity protocol, where there are no streams but the same two threads
        self.threaded = self.cuda if threaded is None else bool(threaded)
        self.cap = None        # the submitting side's capacity: written by submit() only
        self._comm_cap = None  # the communication side's: written by _collect() only (the thread, when there is one)
        self.calls = self.collectives = self.regrows = 0
        self.slab_bytes = 0
        self.slots = [dict(pending=False) for _ in range(2)]
        self.results = [None, None]  # per slot: (all_counts, offs, slab_bytes) of the last merged batch
        self._error = None
        self._thread = None
        self._world = None
        # An optional gate in front of every MERGE (threaded mode): gate_arm() is called by submit() and returns a token,
        # gate_wait(token) by the communication thread, repeatedly, until it returns True or drain() has opened the gates.
        # bench.py arms it with "the next launch of the chain kernels" (OverlapContext.wait_chain_launch): the merge of step k
        # (bound by HBM) then runs beside the issue-bound chain stage of step k + 1 instead of beside its memory-bound index
        # build and candidate scan.  The all-gather is not held: over xGMI it is bound by the links, takes most of a step at
        # 8 ranks and hardly touches HBM.
        self.gate_arm = self.gate_wait = None
        self._gates_open = False
        if self.threaded:
            import queue
            import threading
            self._queue = queue.Queue()
            self._thread = threading.Thread(target=self._comm_loop, name="msgpu-exchange", daemon=True)
            self._thread.start()

    # ---- layout --------------------------------------------------------------------------------------------------
    def _layout(self, cap):
        offs, size = slab_layout(cap, self.wire)
        return tuple(HEADER + o for o in offs), HEADER + size

    def _agree(self, counts):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        mine = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=self.device)
        allc = torch.empty(world * 3, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        self.collectives += 1
        self.cap = self._comm_cap = tuple(int(n * self.slack) + 64 for n in allc.cpu().numpy().reshape(world, 3).max(axis=0))

    def _buffers(self, slot, world, cap):
        """the slot's slab / gathered / header tensors for capacity `cap` (persistent: no allocation per batch).  Called by
        the side that owns the slot at the moment: submit() until the slot is queued, the communication side afterwards."""
        import torch
        if slot.get("layout_cap") == cap and slot.get("world") == world and "slab_bytes" in slot:
            return slot["layout"]  # (the per-batch path: nothing to compute, nothing to allocate)
        offs, slab_bytes = self._layout(cap)
        if slot.get("slab_bytes") != slab_bytes or slot.get("world") != world:
            slot["slab"] = torch.empty(slab_bytes, dtype=torch.uint8, device=self.device)
            slot["gathered"] = torch.empty(world * slab_bytes, dtype=torch.uint8, device=self.device)
            slot["heads_dev"] = torch.empty(world * 3, dtype=torch.int64, device=self.device)
            slot["heads"] = torch.empty(world * 3, dtype=torch.int64, pin_memory=self.cuda)
            slot["hdr"] = torch.empty(3, dtype=torch.int64, pin_memory=self.cuda)  # (pinned: the header copy never blocks)
            slot["slab_bytes"], slot["world"] = slab_bytes, world
        self.slab_bytes = slab_bytes  # (informational: the last layout made)
        slot["layout_cap"], slot["layout"] = cap, (offs, slab_bytes)
        return offs, slab_bytes

    def _gather(self, slot, world):
        """all-gather of the slot's slab + its headers to pinned host memory, on the communication stream (behind the
        event that says the slab is filled)"""
        import torch
        import torch.distributed as dist

        def run():
            slot["hdr"].copy_(torch.tensor(slot["counts"], dtype=torch.int64))
            slot["slab"][:24].view(torch.int64).copy_(slot["hdr"], non_blocking=True)
            dist.all_gather_into_tensor(slot["gathered"], slot["slab"], group=self.group)  # the one collective
            slot["heads_dev"].copy_(slot["gathered"].view(world, slot["slab_bytes"])[:, :24].contiguous().view(torch.int64).view(-1))
            slot["heads"].copy_(slot["heads_dev"], non_blocking=True)
        if self.cuda:
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(slot["filled"])
                run()
                if "done" not in slot:
                    slot["done"] = torch.cuda.Event()
                slot["done"].record(self.comm)
        else:
            run()
        self.collectives += 1

    def _mark_filled(self, slot):
        import torch
        if self.cuda:
            if "filled" not in slot:
                slot["filled"] = torch.cuda.Event()
            slot["filled"].record(torch.cuda.current_stream(self.device))

    # ---- the three calls -----------------------------------------------------------------------------------------
    def submit(self, counts, fill_slab):
        """counts = (n_edges, n_orders, n_ids) of this rank's batch; fill_slab(slab, offs) copies its three tables to the
        byte offsets offs (enqueued on the caller's current stream)."""
        import torch
        import torch.distributed as dist
        if self._world is None:
            self._world = dist.get_world_size(self.group)
        world = self._world
        self._raise()
        if self.cap is None:
            self._agree(counts)
        slot = self.slots[self.calls % 2]
        self.calls += 1
        if slot["pending"]:
            self._finish(slot)  # the batch two submissions ago (collect() after every submit() has done it already)
        if slot.get("new_cap") is not None:  # the fixed point: what the batch two submissions ago published, nothing later
            self.cap = tuple(max(a, b) for a, b in zip(self.cap, slot["new_cap"]))
            slot["new_cap"] = None
        cap = self.cap  # read once: layout, the slot's record and the fit test below see the same capacity
        offs, slab_bytes = self._buffers(slot, world, cap)
        if self.cuda and slot.get("merged") is not None:
            torch.cuda.current_stream(self.device).wait_event(slot["merged"])  # the slab's last reader: two batches ago
        counts = tuple(int(c) for c in counts)
        # (the header is written behind the fill, on the communication stream, by whoever issues the all-gather: the thread
        # that drives the compute stream does nothing here but enqueue the fill and one event)
        slot["counts"], slot["offs"], slot["stash"], slot["cap"] = counts, offs, None, cap
        if all(c <= k for c, k in zip(counts, cap)):
            fill_slab(slot["slab"], offs)
        else:  # outgrown: header only; the tables wait in a private block laid out for their own size
            s_offs, s_size = slab_layout(counts, self.wire)
            stash = torch.empty(HEADER + s_size, dtype=torch.uint8, device=self.device)
            s_offs = tuple(HEADER + o for o in s_offs)
            fill_slab(stash, s_offs)
            slot["stash"] = (stash, s_offs)
        self._mark_filled(slot)
        slot["gate"] = self.gate_arm() if (self.threaded and self.gate_arm is not None) else None
        slot["pending"] = True
        if self.threaded:
            import threading
            slot["finished"] = threading.Event()
            self._queue.put(slot)
        else:
            self._gather(slot, world)

    def _comm_loop(self):
        """the communication thread: all-gather, header check (+ repeat) and merge of every submitted batch, in order"""
        import torch
        import torch.distributed as dist
        if self.cuda:
            torch.cuda.set_device(self.device)
        while True:
            slot = self._queue.get()
            if slot is None:
                return
            try:
                if self._error is None:
                    self._gather(slot, dist.get_world_size(self.group))
                    self._collect(slot)
            except BaseException as exc:  # noqa: BLE001 -- handed to the submitting thread
                self._error = exc
            finally:
                slot["pending"] = False
                slot["finished"].set()

    def _raise(self):
        if self._error is not None:
            exc, self._error = self._error, None
            raise exc

    def _finish(self, slot):
        if self.threaded:
            slot["finished"].wait()
            self._raise()
            return self.results[0 if slot is self.slots[0] else 1][0]
        return self._collect(slot)

    def _relayout(self, slot, world):
        """the capacity grew: this batch's own tables move into a slab of the new layout"""
        import torch
        src, s_offs = slot["stash"] if slot["stash"] is not None else (slot["slab"], slot["offs"])
        counts = slot["counts"]
        slot.pop("slab_bytes", None)
        offs, _ = self._buffers(slot, world, self._comm_cap)
        for nb, so, do in zip(block_bytes(counts, self.wire), s_offs, offs):  # (a block is packed for the rank's own counts)
            slot["slab"][do: do + nb].copy_(src[so: so + nb])
        if self.cuda:
            # `src` (the stash, or the slot's old slab) was allocated on the caller's compute stream and is dropped here while
            # the copies above are still queued on the communication stream: tell the caching allocator about that reader
            src.record_stream(torch.cuda.current_stream(self.device))
        slot["offs"], slot["stash"], slot["cap"] = offs, None, self._comm_cap

    def _collect(self, slot):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        while True:
            if self.cuda:
                slot["done"].synchronize()
            heads = slot["heads"].numpy().reshape(world, 3).astype(np.int64)
            # against the capacity THIS batch was sent with: the batch after it may have gone out before the capacity grew
            if (heads <= np.asarray(slot["cap"], dtype=np.int64)[None, :]).all():
                break
            # same on every rank: a function of gathered headers, in batch order
            self._comm_cap = tuple(max(c, int(n * self.slack) + 64) for c, n in zip(self._comm_cap, heads.max(axis=0)))
            self.regrows += 1
            if self.cuda:  # the re-laid slab is written on the communication stream, behind the failed all-gather
                with torch.cuda.stream(self.comm):
                    self._relayout(slot, world)
                    self._mark_filled(slot)
            else:
                self._relayout(slot, world)
            self._gather(slot, world)
        k = 0 if slot is self.slots[0] else 1
        if self.threaded and slot.get("gate") is not None and self.gate_wait is not None:
            # the all-gather (bound by the links: it hardly touches HBM) went out at once; the merge (bound by HBM) waits here
            while not self._gates_open and not self.gate_wait(slot["gate"]):
                pass  # (gate_wait times out every millisecond or so: drain() is noticed)
        if self.cuda:
            with torch.cuda.stream(self.comm):
                self.merge(slot["gathered"], heads, slot["offs"], slot["slab_bytes"], k, self.comm)
                if slot.get("merged") is None:
                    slot["merged"] = torch.cuda.Event()
                slot["merged"].record(self.comm)
        else:
            self.merge(slot["gathered"], heads, slot["offs"], slot["slab_bytes"], k, None)
        self.results[k] = (heads, slot["offs"], slot["slab_bytes"])
        slot["new_cap"] = self._comm_cap  # adopted by the submit() that takes this slot next
        slot["pending"] = False
        return heads

    def collect(self):
        """without a communication thread: finish the batch BEFORE the one just submitted (no-op when there is none) -> its
        all_counts or None.  With one: nothing to do here (returns None)."""
        if self.threaded:
            self._raise()
            return None
        slot = self.slots[self.calls % 2]  # the slot the NEXT submit would take = the older of the two
        return self._collect(slot) if slot["pending"] else None

    def drain(self):
        """finish everything submitted; returns the all_counts of the last batch"""
        last = None
        self._gates_open = True  # nothing comes behind these batches: they go now
        for k in (self.calls % 2, (self.calls + 1) % 2):  # older first
            if self.slots[k]["pending"] or (self.threaded and "finished" in self.slots[k]):
                r = self._finish(self.slots[k])
                last = r if r is not None else last
        if self.cuda:
            self.comm.synchronize()
        self._gates_open = False
        return last

    def close(self):
        if self._thread is not None:
            # a batch still held at its gate (close() without drain(): the caller's step raised) goes now -- the gate is a
            # scheduling hint, and a communication thread looping in gate_wait would never reach the None behind its batch
            self._gates_open = True
            self._queue.put(None)
            self._thread.join()
            self._thread = None
