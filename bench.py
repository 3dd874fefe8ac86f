#!/usr/bin/env python3
"""bench.py -- overlap hot path of MuCHSALSA on MI355X: overlap-pairs/s on the BASELINE.json workload.

A step = one pass of the hot path over the synthetic batch, starting with the accepted-row table resident in HBM:
  msgpu_load_rows_device   (device index build = MatchMap/Graph-vertex fill)
  msgpu_calculate_edges    (MatchMap::calculateEdges)
  msgpu_chaining_and_overlaps (the chainingAndOverlaps fan-out)
  N > 1: one RCCL all-gather of the per-rank edge + order + id tables (each rank owns the edges with v1 % N == rank)

Contract: python bench.py --gpus N --steps K --warmup W ; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE),
# collected with rocprofv3 in separate --pmc passes of this same command (tools/pmc_passes.sh):
# profiles/r1_08_final/pmc_summary.csv for the three chain kernels together, profiles/r1_05_gather for the gather.
# They cannot be read from inside this process, so they are quoted for the one configuration they were measured on.
PMC_TRAFFIC_BYTES = {("cfg3", 1): {"k_chain": 5.35e9, "k_gather_packed": 1.48e9}}

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "cfg3": dict(n_reads=100_000, read_len=10_000, n_anchors=500_000, seed=43,
                 name="100k synthetic Nanopore x10 kb, 500k unitig anchors, 10x coverage (BASELINE.json configs[2])"),
    # BASELINE.json configs[1]
    "cfg2": dict(n_reads=10_000, read_len=5_000, n_anchors=50_000, seed=42,
                 name="10k synthetic Nanopore x5 kb, 50k unitig anchors, 10x coverage (BASELINE.json configs[1])"),
    "tiny": dict(n_reads=2_000, read_len=5_000, n_anchors=10_000, seed=7, name="2k x5 kb, 10k anchors (smoke)"),
}


def _cpu_sample(job):
    """One sample of the workload shape through the C oracle on one core (runs in a worker process: no torch, no GPU)."""
    n_reads, read_len, n_anchors, seed = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ms_oracle_ctypes as oracle
    from muchsalsa_amd import synth
    rows = synth.synth_rows(n_reads, read_len, n_anchors, seed)
    t0 = time.perf_counter()
    t = oracle.overlap(rows)
    dt = time.perf_counter() - t0
    return len(t["edges"]), len(t["ems"]), int(t["compat_checks"]), dt


def cpu_baseline(workload, budget_reads, cores):
    """The CPU side of the comparison: the C oracle (a single-thread restatement of the reference's algorithm) on a
    bounded sample of the same workload shape -- once on one core, and once as `cores` independent samples (different
    seeds), one per core, at the same time.  The second figure is what a perfectly scaling multi-threaded CPU build
    could reach on this host; the reference's own ThreadPool fan-out scales worse than that."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ms_oracle_ctypes as oracle
    oracle.build()
    w = WORKLOADS[workload]
    scale = min(1.0, budget_reads / w["n_reads"])
    n_reads = max(1, int(w["n_reads"] * scale))
    n_anchors = max(1, int(w["n_anchors"] * scale))
    e1, m1, c1, dt1 = _cpu_sample((n_reads, w["read_len"], n_anchors, w["seed"]))
    out = {
        "value": e1 / dt1, "unit": "overlap-pairs/s", "cores": 1, "kind": "port",
        "sample": "%d reads x %d bp, %d anchors (same generator/seed/density as the GPU workload, %.0f%% scale): "
                  "%d edges, %d EdgeMatches, %d compat checks in %.2f s" % (
                      n_reads, w["read_len"], n_anchors, 100 * scale, e1, m1, c1, dt1),
    }
    if cores > 1:
        # worker processes are SPAWNED (this process has initialised the GPU; its children must not inherit that)
        jobs = [(n_reads, w["read_len"], n_anchors, w["seed"] + 1000 + k) for k in range(cores)]
        t0 = time.perf_counter()
        try:
            with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as pool:
                res = list(pool.map(_cpu_sample, jobs))
        except Exception as exc:  # a box that refuses worker processes must not cost the whole bench line
            out["sample"] += "; the %d-core leg failed (%s: %s), one core reported" % (cores, type(exc).__name__, exc)
            return out
        wall = time.perf_counter() - t0
        busy = max(r[3] for r in res)
        out.update({
            "one_core_value": out["value"], "value": sum(r[0] for r in res) / busy, "cores": cores,
            "sample": "%d independent samples of that shape (seeds differ), one per core, run together: %d edges in "
                      "%.2f s (slowest worker; %.2f s with process start-up); one core alone: %s" % (
                          cores, sum(r[0] for r in res), busy, wall, out["sample"]),
        })
    return out


def consensus_leg(torch, dev, world, rank, w, steps, warmup):
    """Device half of the consensus stage (A9) on the same synthetic reads: slice / reverse-complement / stitch.

    Sequences: a random genome; read i = genome[start_i : start_i + L], reverse-complemented for '-' reads (built on the
    device with the gather kernel itself and installed as the nanopore store).  Layout (host, numpy, NOT timed -- the
    reference's layout logic, assemblePath's anchor DAG, is not built yet): reads in genome order stitched by the
    updateConsensusBase append rule (ap.cpp:205-229: a read contributes the part that extends the contig) -> target
    contigs; every read, oriented to the genome strand, is a query.  Timed: the gather kernel producing target+queries.
    Checked at full size: the stitched target equals the genome intervals it covers, byte for byte.
    """
    from muchsalsa_amd import sequences as S, synth
    from muchsalsa_amd._lib import COPY_DTYPE, COPY_ILLUMINA, COPY_REVCOMP
    n_reads, L = w["n_reads"], w["read_len"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, w["seed"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(w["seed"])
    genome = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[
        torch.randint(0, 4, (G,), device=dev, generator=gen)]
    store = S.SeqStore(device=dev.index)
    tstream = torch.cuda.Stream(device=dev)  # a real (non-null) stream: kernels and timing events share it
    torch.cuda.synchronize()
    stream = tstream.cuda_stream
    store.upload_device(S.ILLUMINA, genome.data_ptr(), G, [0], [G])  # the genome as "unitig 0"

    # reads on the device: one piece per read out of the genome
    mk = np.zeros(n_reads, dtype=COPY_DTYPE)
    mk["src_off"], mk["dst_off"], mk["len"] = r_start, np.arange(n_reads, dtype=np.uint64) * L, L
    mk["flags"] = COPY_ILLUMINA | np.where(r_fwd, 0, COPY_REVCOMP).astype(np.uint32)
    d_reads = torch.empty(n_reads * L, dtype=torch.uint8, device=dev)
    store.run(store.plan(mk), d_reads.data_ptr(), d_reads.numel(), stream=stream)
    torch.cuda.synchronize()
    store.upload_device(S.NANOPORE, d_reads.data_ptr(), d_reads.numel(), np.arange(n_reads, dtype=np.uint64) * L,
                        np.full(n_reads, L, dtype=np.uint64))
    del d_reads
    store.pack()  # 2 bits per base in HBM (+ exception list, empty here): the form the timed gather reads

    # layout: append rule over reads in genome order
    order = np.argsort(r_start, kind="stable")
    st, en = r_start[order], r_start[order] + L
    cur = np.maximum.accumulate(np.concatenate([[0], en[:-1]]))  # contig end before each read
    lo = np.maximum(st, cur)                                      # a gap (st > cur) starts a new contig
    ext = en > cur
    rid, lo, hi, fwd = order[ext], lo[ext], en[ext], r_fwd[order][ext]
    t_len = (hi - lo).astype(np.uint64)
    t_off = np.concatenate([[0], np.cumsum(t_len)[:-1]]).astype(np.uint64)
    T = int(t_len.sum())
    tgt = np.zeros(len(rid), dtype=COPY_DTYPE)
    # genome interval [lo, hi) of read r: read coordinates lo-start.. (forward) / start+L-hi.. (reverse strand)
    left = np.where(fwd, lo - r_start[rid], r_start[rid] + L - hi)
    tgt["src_off"], tgt["dst_off"], tgt["len"] = rid.astype(np.uint64) * L + left.astype(np.uint64), t_off, t_len
    tgt["flags"] = np.where(fwd, 0, COPY_REVCOMP).astype(np.uint32)
    qry = np.zeros(n_reads, dtype=COPY_DTYPE)
    qry["src_off"] = np.arange(n_reads, dtype=np.uint64) * L
    qry["dst_off"] = np.uint64(T) + np.arange(n_reads, dtype=np.uint64) * L
    qry["len"], qry["flags"] = L, np.where(r_fwd, 0, COPY_REVCOMP).astype(np.uint32)
    pieces = np.concatenate([tgt, qry])
    mine = pieces[rank::world] if world > 1 else pieces
    plan = store.plan(mine)
    out_bytes = T + n_reads * L
    out = torch.empty(out_bytes, dtype=torch.uint8, device=dev)

    for _ in range(warmup):
        store.run(plan, out.data_ptr(), out_bytes, stream=stream)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(tstream)
    for _ in range(steps):
        store.run(plan, out.data_ptr(), out_bytes, stream=stream)
    ev1.record(tstream)
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0) / steps
    ms = ev0.elapsed_time(ev1) / steps
    assert ms > 0.5 * wall_ms - 0.05, "event timing (%.4f ms) disagrees with the wall clock (%.4f ms)" % (ms, wall_ms)

    # full-size check: the stitched target equals the genome it covers; queries equal the genome under each read
    exp = np.zeros(len(rid), dtype=COPY_DTYPE)
    exp["src_off"], exp["dst_off"], exp["len"], exp["flags"] = lo.astype(np.uint64), t_off, t_len, COPY_ILLUMINA
    want = torch.empty(max(T, 1), dtype=torch.uint8, device=dev)
    store.run(store.plan(exp), want.data_ptr(), T, stream=stream)
    torch.cuda.synchronize()
    ok = True
    if world == 1:
        ok = bool(torch.equal(out[:T], want[:T]))
        probe = np.random.default_rng(0).choice(n_reads, 64, replace=False)
        for i in probe:
            a = int(r_start[i])
            ok &= bool(torch.equal(out[T + int(i) * L: T + (int(i) + 1) * L], genome[a:a + L]))
    bases_mine = int(mine["len"].sum())
    store.close()
    return {"ms": ms, "target_bases": T, "query_bases": n_reads * L, "pieces": int(len(pieces)),
            "bases_this_rank": bases_mine, "verified": ok}


def assemble_leg(torch, dev, w, rows, read_names, anchor_names, tables, window_mb):
    """assemblePath (A9) end to end on a bounded sample: every read starting in the first `window_mb` Mb of the synthetic
    genome is chained into paths (muchsalsa_amd.synth.chain_paths, the stand-in for linearizeGraph) over the overlap
    tables the timed steps just produced; then, timed: host layout of every path (msgpu_assembly_add_path) and ONE
    gather + FASTA-wrapping pass on the device with the texts copied back (msgpu_assembly_finish).  Self-check with the
    product's own meter: the banded edit distance of every query record against the stretch of its contig that its PAF
    line names (msgpu_assembly_validate).  Byte parity with the restatement of ap.cpp is the tests' job
    (tests/test_gpu_assemble.py); nothing under oracle/ is touched here."""
    from muchsalsa_amd import sequences as S, synth
    from muchsalsa_amd._lib import COPY_DTYPE, COPY_ILLUMINA, COPY_REVCOMP
    from muchsalsa_amd.assembly import Assembly
    n_reads, L, seed = w["n_reads"], w["read_len"], w["seed"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, seed)
    a_start, a_len = synth.anchor_layout(n_reads, L, w["n_anchors"], seed)
    read_orig = np.array([int(n[1:]) for n in read_names])      # Registry id -> generator index
    anchor_orig = np.array([int(n[1:]) for n in anchor_names])
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    genome = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[
        torch.randint(0, 4, (G,), device=dev, generator=gen)]
    store = S.SeqStore(device=dev.index)
    store.upload_device(S.ILLUMINA, genome.data_ptr(), G, [0], [G])
    # the two stores, keyed by Registry id, cut out of the genome on the device
    rs, rf = r_start[read_orig], r_fwd[read_orig]
    mk = np.zeros(len(read_orig), dtype=COPY_DTYPE)
    mk["src_off"], mk["dst_off"], mk["len"] = rs, np.arange(len(rs), dtype=np.uint64) * L, L
    mk["flags"] = COPY_ILLUMINA | np.where(rf, 0, COPY_REVCOMP).astype(np.uint32)
    d_reads = torch.empty(len(rs) * L, dtype=torch.uint8, device=dev)
    store.run(store.plan(mk), d_reads.data_ptr(), d_reads.numel())
    al = a_len[anchor_orig].astype(np.uint64)
    aoff = np.concatenate([[0], np.cumsum(al)[:-1]]).astype(np.uint64)
    mk = np.zeros(len(anchor_orig), dtype=COPY_DTYPE)
    mk["src_off"], mk["dst_off"], mk["len"], mk["flags"] = a_start[anchor_orig], aoff, al, COPY_ILLUMINA
    d_anch = torch.empty(int(al.sum()), dtype=torch.uint8, device=dev)
    store.run(store.plan(mk), d_anch.data_ptr(), d_anch.numel())
    store.synchronize()
    store.upload_device(S.NANOPORE, d_reads.data_ptr(), d_reads.numel(), np.arange(len(rs), dtype=np.uint64) * L,
                        np.full(len(rs), L, dtype=np.uint64))
    store.upload_device(S.ILLUMINA, d_anch.data_ptr(), d_anch.numel(), aoff, al)
    del d_reads, d_anch
    store.pack()

    t0 = time.perf_counter()
    paths = synth.chain_paths(tables, rs, rf, L, int(window_mb * 1e6), max_reads=12)
    t_paths = time.perf_counter() - t0
    prepared = [Assembly.prepare(p, st, None, None, i) for i, (p, st) in enumerate(paths)]
    threads = max(1, min(16, os.cpu_count() or 1))
    warm = Assembly(store)  # warm-up pass (untimed), like the W warm-up steps of the overlap half: first-touch
    warm.set_rows(rows)     # allocations, page pinning, code-object load
    warm.add_prepared_batch(prepared, threads)
    warm.finish()
    warm.close()
    # Host-side times on a shared box are noisy (a scheduler hiccup once turned 1 ms of layout into 8): the leg is run
    # on REPS fresh assemblies and the medians are reported, with every sample listed beside them.
    REPS, samples, asm = 5, [], None
    for _ in range(REPS):
        if asm is not None:
            asm.close()
        asm = Assembly(store)
        t0 = time.perf_counter()
        asm.set_rows(rows)
        t_i = time.perf_counter() - t0
        t0 = time.perf_counter()
        status = asm.add_prepared_batch(prepared, threads)
        t_l = time.perf_counter() - t0
        assert not status.any(), "a synthetic chain was rejected: %r" % status
        t0 = time.perf_counter()
        asm.finish()
        t_d = time.perf_counter() - t0
        samples.append((t_l, t_d, t_i))
    t_layout, t_device, t_index = (float(np.median([x[k] for x in samples])) for k in range(3))
    info, qinfo = asm.paths, asm.queries
    T, Q = int(info["target_len"].sum()), int(qinfo["len"].sum())
    # A10: the banded anti-diagonal DP kernel as the assembly's self-check (every query against its PAF window)
    band = 64
    asm.validate(band)  # warm-up
    t0 = time.perf_counter()
    dist, cells = asm.validate(band)
    t_val = time.perf_counter() - t0

    res = {"paths": len(paths), "reads_on_paths": int(sum(len(p) for p, _ in paths)), "target_bases": T,
            "query_bases": Q, "queries": int(len(qinfo)), "pieces": int(len(asm.pieces)),
            "layout_ms": 1e3 * t_layout, "layout_threads": threads, "device_ms": 1e3 * t_device, "row_index_ms": 1e3 * t_index,
            "timing": "medians of %d fresh assemblies" % REPS,
            "layout_ms_samples": [round(1e3 * x[0], 3) for x in samples],
            "device_ms_samples": [round(1e3 * x[1], 3) for x in samples],
            "path_builder_ms_untimed": 1e3 * t_paths,
            "text_bytes": len(asm.text(0)) + len(asm.text(1)) + len(asm.text(2)), "window_mb": window_mb,
            "validate": {"kernel": "k_edit_distance", "band": band, "pairs": int(len(dist)), "ms_incl_copies": 1e3 * t_val,
                         "dp_cells": cells, "dp_gcells_per_s": cells / t_val / 1e9,
                         "queries_within_band": int((dist <= band).sum()),
                         "median_distance": float(np.median(dist)) if len(dist) else None}}
    store.close()  # closes the assembly laid out over it as well
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample-reads", type=int, default=25_000,
                    help="reads in the CPU-baseline sample (0 disables the baseline leg)")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="host cores of the CPU-baseline leg (0 = min(16, cpu count): the box's CPU share per GPU)")
    ap.add_argument("--no-consensus", action="store_true", help="skip the consensus (sequence gather) leg")
    ap.add_argument("--assemble-window-mb", type=float, default=10.0,
                    help="assemblePath leg: chain the reads starting in the first this-many Mb of the genome (0 = skip)")
    ap.add_argument("--graph-stage", action="store_true",
                    help="also time findContractionEdges (GPU) and the host graph stage on the job's tables (slow on "
                         "this workload: one 100k-vertex component, see DESIGN.md section 10)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 exchange path (all-gather + merge) even at world size 1 (used by the GPU tests)")
    args = ap.parse_args()

    # stdout must carry exactly one JSON line: libraries (RCCL prints a version banner on stdout) write to fd 1 while
    # we run, so fd 1 is pointed at stderr for the duration and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from muchsalsa_amd import distributed as D, overlap, synth
    from muchsalsa_amd._lib import EDGE_DTYPE, ORDER_DTYPE

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    w = WORKLOADS[args.workload]
    rows, read_names, anchor_names = synth.accepted_rows(
        synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).to(dev)  # the accepted-row table, resident in HBM
    torch.cuda.synchronize()

    ctx = overlap.OverlapContext(device=local_rank)
    # One real (non-null) stream carries everything: libmsgpu's kernels, torch's allocations and the RCCL collective
    # are ordered by it, so the all-gather cannot overtake the table copies nor the merge kernel the all-gather.
    work = torch.cuda.Stream(device=dev)
    ctx.set_stream(work.cuda_stream)
    if world > 1:
        ctx.set_shard(rank, world)
    ctx.set_id_space(len(read_names), len(anchor_names))  # Registry sizes, known to whoever parsed the PAF
    merged_keep = {}

    def step():
        ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        c = ctx.counts()
        if multi:
            # merge the edge list: ONE all-gather of the per-rank (edges | orders | ids) slab over xGMI, then the
            # HIP compaction/re-base kernel (msgpu_merge_gathered)
            def fill(slab, offs):
                ctx.copy_tables_device(d_edges=slab.data_ptr() + offs[0], d_orders=slab.data_ptr() + offs[1],
                                       d_ids=slab.data_ptr() + offs[2])
            gathered, allc, offs, slab_bytes = D.gather_slabs((c.n_edges, c.n_orders, c.n_ids), fill, dev)
            tot = allc.sum(axis=0)
            m_e = torch.empty(max(int(tot[0]), 1) * EDGE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            m_o = torch.empty(max(int(tot[1]), 1) * ORDER_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            m_i = torch.empty(max(int(tot[2]), 1) * 4, dtype=torch.uint8, device=dev)
            ctx.merge_gathered(gathered.data_ptr(), allc, slab_bytes, offs, m_e.data_ptr(), m_o.data_ptr(),
                               m_i.data_ptr())
            merged_keep.update(e=m_e, o=m_o, i=m_i, tot=tot)
            return c, allc
        return c, None

    with torch.cuda.stream(work):
        for _ in range(args.warmup):
            step()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chain_ms = []
        for _ in range(args.steps):
            c, allc = step()
            chain_ms.append(ctx.timings().chain_kernel_ms)  # HIP events on the launch stream (syncs that stream only)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt = time.perf_counter() - t0
    tm = ctx.timings()
    merge_ok = None
    if multi and rank == 0:  # not timed: the merged edge list must be a consistent table (and, alone, equal our own)
        tot = merged_keep["tot"]
        me = merged_keep["e"].cpu().numpy()[: int(tot[0]) * EDGE_DTYPE.itemsize].view(EDGE_DTYPE)
        mo = merged_keep["o"].cpu().numpy()[: int(tot[1]) * ORDER_DTYPE.itemsize].view(ORDER_DTYPE)
        mi = merged_keep["i"].cpu().numpy()[: int(tot[2]) * 4].view("<u4")
        merge_ok = bool(
            np.array_equal(me["order_off"], np.concatenate([[0], np.cumsum(me["order_cnt"])[:-1]]).astype(np.uint64))
            and np.array_equal(mo["edge_idx"], np.repeat(np.arange(len(me), dtype=np.uint32), me["order_cnt"]))
            and np.array_equal(mo["ids_off"], np.concatenate([[0], np.cumsum(mo["ids_cnt"])[:-1]]).astype(np.uint64))
            and int(mo["ids_cnt"].sum()) == len(mi) and bool(np.all(mo["base"] == me["v1"][mo["edge_idx"]])))
        if world == 1:
            own = ctx.tables()
            merge_ok = merge_ok and me.tobytes() == own["edges"].tobytes() and mo.tobytes() == own["orders"].tobytes() \
                and mi.tobytes() == own["ids"].tobytes()
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        n_edges_total = int(allc[:, 0].sum())
    else:
        n_edges_total = int(c.n_edges)

    asm_leg = graph_leg = None
    def aux_legs():
        nonlocal asm_leg, graph_leg
        # what follows the chaining fan-out in main(): findContractionEdges on the GPU (tables still resident), then the
        # host graph stage (src/main.cpp:183-310).  Reported beside the metric, never inside `value`.
        torch.cuda.synchronize()
        ctx.find_contraction_edges()  # warm-up: arena allocation
        t0 = time.perf_counter()
        contraction = ctx.find_contraction_edges()
        t_contr = time.perf_counter() - t0
        tables = ctx.tables()
        graph_leg = {"find_contraction_edges_ms_incl_copy_back": 1e3 * t_contr,
                     "contraction_edges": int((contraction >= 0).sum()),
                     "shadow_edges": int(tables["edges"]["shadow"].sum())}
        if args.graph_stage:
            from muchsalsa_amd.graph import GraphStage
            read_len, read_first = ctx.reads()
            t0 = time.perf_counter()
            gs = GraphStage(tables, read_len, read_first)
            t_build = time.perf_counter() - t0
            t0 = time.perf_counter()
            gs.clean_up(contraction, None)
            t_clean = time.perf_counter() - t0
            t0 = time.perf_counter()
            gs.linearize()
            t_lin = time.perf_counter() - t0
            st = gs.stats
            graph_leg.update({"graph_build_ms": 1e3 * t_build, "clean_up_ms": 1e3 * t_clean, "linearize_ms": 1e3 * t_lin,
                              "vertices_after": int(st.n_vertices), "edges_after": int(st.n_edges),
                              "decycled_edges": int(st.n_decycled_edges), "components": int(st.n_components),
                              "paths": int(st.n_paths), "path_reads": int(st.n_path_reads)})
            gs.close()
        ctx.close()
        asm_leg = assemble_leg(torch, dev, w, rows, read_names, anchor_names, tables, args.assemble_window_mb)
        del tables

    # the legs reported BESIDE the metric must never cost the metric line itself
    if world == 1 and rank == 0 and args.assemble_window_mb > 0 and not args.no_consensus:
        try:
            aux_legs()
        except Exception as exc:  # noqa: BLE001
            asm_leg = {"error": "%s: %s" % (type(exc).__name__, exc)}
    cons = None
    if not args.no_consensus:
        ctx.close()  # give the arena back before the ~3 GB of sequence buffers
        try:
            cons = consensus_leg(torch, dev, world, rank, w, args.steps, args.warmup)
        except Exception as exc:  # noqa: BLE001
            if multi:
                raise  # the other ranks are waiting in the all-reduce below
            cons = None
            consensus_error = "%s: %s" % (type(exc).__name__, exc)
        else:
            consensus_error = None
    else:
        consensus_error = None
    if cons is not None:
        if multi:
            tt = torch.tensor([cons["ms"]], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cons["ms"] = float(tt.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        # algorithmic bytes of the dominant kernel (k_chain), SURVEY.md section 8(d):
        #   96 B per EdgeMatch (32 B EdgeMatch written + two 32 B VertexMatch rows read) + 64 B per order + 4 B per id
        alg_bytes = 96 * c.n_ems + 64 * c.n_orders + 4 * c.n_ids
        k_ms = float(np.mean(chain_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "overlap-pairs/s", "value": n_edges_total / (dt / args.steps), "unit": "overlap-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32+f64",
            "data": "synthetic",
            "config": {"workload": w["name"], "rows": int(len(rows)), "reads": int(c.n_reads),
                       "anchors": int(c.n_anchors), "edges": n_edges_total, "edgematches_rank0": int(c.n_ems),
                       "orders_rank0": int(c.n_orders), "parallelism": "edges sharded by v1 %% %d" % world,
                       "edges_proven_clean_rank0": int(c.n_edges_fastpath), "merged_edge_list_consistent": merge_ok,
                       "note": "value = overlap half of the metric; the consensus half is reported under 'consensus' "
                               "(the gather kernel alone at full size) and 'assemble_path' (assemblePath end to end: "
                               "host layout + gather + FASTA wrapping, on a bounded sample of paths)"},
            "stage_ms": {"index": tm.index_ms, "candidates": tm.candidates_ms, "chain_total": tm.chain_ms,
                         "chain_kernel": k_ms, "compact": tm.compact_ms},
            "roofline": {"bound": "hbm", "kernel": "k_chain + k_chain_sub<32> + k_chain_sub<16> + k_chain_sub<8> (one pass "
                                                   "over the edges, four launches by edge size)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": PMC_TRAFFIC_BYTES.get((args.workload, world), {}).get("k_chain"),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": k_ms,
                         "note": "kernel_ms = the four launches together (HIP events around them on the launch stream); "
                                 "vector-ALU bound: 0.75-0.92 of the issue cycles; traffic is twice the algorithmic bytes because the "
                                 "edges run in size order, not table order (rows of a read are re-fetched): "
                                 "profiles/r1_08_final/README.md"},
        }
        if cons is not None:
            # algorithmic bytes on the 2-bit store: 0.25 B read + 1 B written per base (SURVEY 8(d) counted 1 B + 1 B for a
            # byte-per-base source; that figure is kept as "bytes_if_byte_store" for comparison)
            gb = 1.25 * (cons["target_bases"] + cons["query_bases"]) / 1e9
            g_gbs = gb / (cons["ms"] * 1e-3)
            out["consensus"] = {
                "stage": "slice / reverse-complement / stitch kernel k_gather_packed on the 2-bit sequence store "
                         "(layout precomputed on the host, not timed)",
                "consensus_mbases_per_s": cons["target_bases"] / (cons["ms"] * 1e-3) / 1e6,
                "query_mbases_per_s": cons["query_bases"] / (cons["ms"] * 1e-3) / 1e6,
                "target_bases": cons["target_bases"], "query_bases": cons["query_bases"], "pieces": cons["pieces"],
                "ms": cons["ms"], "verified_against_genome": cons["verified"],
                "roofline": {"bound": "hbm", "kernel": "k_gather_packed", "achieved": g_gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": g_gbs / HBM_PEAK_GBS,
                             "traffic": PMC_TRAFFIC_BYTES.get((args.workload, world), {}).get("k_gather_packed"),
                             "algorithmic_bytes_per_launch": int(gb * 1e9),
                             "bytes_if_byte_store": int(2 * (cons["target_bases"] + cons["query_bases"])),
                             "note": "2 bits in + 1 byte out per base; the same launch on a byte-per-base source would "
                                     "move 2 B per base"},
            }
        if graph_leg is not None:
            out["graph_stage"] = graph_leg
        if consensus_error is not None:
            out["consensus"] = {"error": consensus_error}
        if asm_leg is not None and "error" in asm_leg:
            out["assemble_path"] = asm_leg
        elif asm_leg is not None:
            tot_ms = asm_leg["layout_ms"] + asm_leg["device_ms"]
            asm_leg["consensus_mbases_per_s"] = asm_leg["target_bases"] / (tot_ms * 1e-3) / 1e6
            asm_leg["stage"] = ("assemblePath on a bounded sample: host layout of every path + one gather + FASTA "
                                "wrapping + copy-back of target.fa/query.fa (layout_ms + device_ms)")
            out["assemble_path"] = asm_leg
        if world == 1 and args.cpu_sample_reads > 0:
            cores = args.cpu_cores if args.cpu_cores > 0 else max(1, min(16, os.cpu_count() or 1))
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_reads, cores)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
