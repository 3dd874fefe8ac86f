#!/usr/bin/env python3
"""bench.py -- overlap hot path of MuCHSALSA on MI355X: overlap-pairs/s on the BASELINE.json workload.

A step = one pass of the hot path over the synthetic batch, starting with the accepted-row table resident in HBM:
  msgpu_load_rows_device   (device index build = MatchMap/Graph-vertex fill)
  msgpu_calculate_edges    (MatchMap::calculateEdges)
  msgpu_chaining_and_overlaps (the chainingAndOverlaps fan-out)
  N > 1: one RCCL all-gather of the per-rank edge + order + id tables (each rank owns the edges with v1 % N == rank)

Contract: python bench.py --gpus N --steps K --warmup W ; rank 0 prints ONE JSON line.  With --gpus N > 1 and no
WORLD_SIZE in the environment the script starts the N ranks itself (torch.distributed.run, before any GPU call) and
relays rank 0's line.

Beside `value` (device-resident in / out, as the contract asks) the line carries:
  roofline       the chain kernels against the HBM peak (algorithmic bytes / HIP-event time; traffic from profiles/)
  host_to_host   SURVEY section 8(d)'s own region: rows in pinned host memory -> all four tables in pinned host memory
                 (msgpu_overlap_batched: batches on two streams, the copy of batch k behind the compute of batch k+1)
  consensus      the gather kernel at full size; assemble_path: assemblePath over chains tiling the WHOLE genome
  graph_stage    findContractionEdges (GPU) + clean-up + linearizeGraph (host, flat CSR) on the job's tables, and
                 assemblePath over the paths it yields
  cpu_baseline   the C oracle (single-thread restatement of the reference) on the box's host cores
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CHAIN_KERNELS = ("msgpu::k_chain", "msgpu::k_chain_sub_all")  # one pass over the edges: 33..64 EdgeMatches | <= 32 (three widths)
CHAIN_KERNELS_BEFORE = ("msgpu::k_chain", "void msgpu::k_chain_sub<32>", "void msgpu::k_chain_sub<16>", "void msgpu::k_chain_sub<8>")

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "cfg3": dict(n_reads=100_000, read_len=10_000, n_anchors=500_000, seed=43,
                 name="100k synthetic Nanopore x10 kb, 500k unitig anchors, 10x coverage (BASELINE.json configs[2])"),
    # BASELINE.json configs[1]
    "cfg2": dict(n_reads=10_000, read_len=5_000, n_anchors=50_000, seed=42,
                 name="10k synthetic Nanopore x5 kb, 50k unitig anchors, 10x coverage (BASELINE.json configs[1])"),
    "tiny": dict(n_reads=2_000, read_len=5_000, n_anchors=10_000, seed=7, name="2k x5 kb, 10k anchors (smoke)"),
}


# ---- HBM traffic of the dominant kernels from the committed counter captures ---------------------------------------------

def pmc_traffic(workload, world, kernels):
    """HBM bytes per launch of `kernels` together, from the newest profiles/*/pmc_summary.csv whose pmc_meta.json names
    this (workload, world): FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, both KiB per
    dispatch.  -> (bytes or None, source / reason, {kernel: VALU issue numbers} or None)"""
    import csv
    best = None
    for meta in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_meta.json"))):
        try:
            m = json.load(open(meta))
        except (OSError, ValueError):
            continue
        f = os.path.join(os.path.dirname(meta), "pmc_summary.csv")
        if m.get("workload") == workload and int(m.get("world", 0)) == world and os.path.exists(f):
            best = f  # sorted(): the last match is the newest round
    if best is None:
        return None, "no counter capture under profiles/ for (%s, %d GPU): run tools/pmc_passes.sh" % (workload, world), None
    table = list(csv.DictReader(open(best)))
    gone = capture_is_stale(table)
    if gone:
        return None, "%s was taken with kernels this build no longer has (%s): traffic dropped" % (
            os.path.relpath(best, ROOT), ", ".join(gone[:4])), None
    rows = {r["kernel"]: r for r in table}
    total, valu = 0.0, {}
    for k in kernels:
        r = rows.get(k)
        if r is None or not r.get("FETCH_SIZE") or not r.get("WRITE_SIZE"):
            return None, "%s lacks FETCH_SIZE/WRITE_SIZE for %s" % (os.path.relpath(best, ROOT), k), None
        total += (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
        if r.get("SQ_INSTS_VALU"):
            valu[k] = float(r["SQ_INSTS_VALU"])
    return total, os.path.relpath(best, ROOT), valu


def current_library_kernels():
    """base names (k_chain, k_index_bin, ...) of the kernels in the built libmsgpu.so, read from its symbol strings"""
    import re
    try:
        data = open(os.path.join(ROOT, "muchsalsa_amd", "libmsgpu.so"), "rb").read()
    except OSError:
        return set()
    return {name[:int(n)].decode() for n, name in re.findall(rb"_ZN5msgpu(\d+)(k_[A-Za-z0-9_]+)", data)}


def capture_is_stale(table):
    """kernels of a counter capture that the current build no longer has (by base name) -> sorted list"""
    import re
    current = current_library_kernels()
    if not current:
        return []
    gone = set()
    for r in table:
        m = re.search(r"msgpu::(k_[A-Za-z0-9_]+)", r["kernel"])
        if m and m.group(1) not in current:
            gone.add(m.group(1))
    return sorted(gone)


def pmc_stage(workload, world, prefixes):
    """Counter traffic of every kernel whose name starts with one of `prefixes` (a stage's kernels), from the same capture as
    pmc_traffic: -> (bytes per step or None, source, [kernels counted])"""
    import csv
    best = None
    for meta in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_meta.json"))):
        try:
            m = json.load(open(meta))
        except (OSError, ValueError):
            continue
        f = os.path.join(os.path.dirname(meta), "pmc_summary.csv")
        if m.get("workload") == workload and int(m.get("world", 0)) == world and os.path.exists(f):
            best = f
    if best is None:
        return None, "no counter capture under profiles/ for (%s, %d GPU)" % (workload, world), []
    total, used = 0.0, []
    table = list(csv.DictReader(open(best)))
    # steps in the capture = dispatches of a kernel that runs once per step
    anchor = [float(r["dispatches"]) for r in table if r["kernel"].replace("void ", "") == "msgpu::k_compact"]
    if not anchor:  # (without the once-per-step kernel the per-step share of a dispatch count is unknown)
        return None, "%s has no msgpu::k_compact row to count its steps by" % os.path.relpath(best, ROOT), []
    steps = max(anchor)
    gone = capture_is_stale(table)  # the capture must be of THIS build's kernels
    if gone:
        return None, "%s was taken with kernels this build no longer has (%s): traffic dropped" % (
            os.path.relpath(best, ROOT), ", ".join(gone[:4])), []
    for r in table:
        name = r["kernel"].replace("void ", "")
        if any(name.startswith(p) for p in prefixes) and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
            # per-dispatch averages x dispatches per step (a kernel launched twice per step counts twice)
            total += (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0 * float(r["dispatches"]) / steps
            used.append(name)
    return (total if used else None), os.path.relpath(best, ROOT), used


STAGE_KERNELS = {  # the kernels of each stage, by name prefix (profiles/*/kernel_stats.csv)
    "index": ("msgpu::k_index_", "msgpu::k_bin_scan", "msgpu::k_sort_read", "msgpu::k_check_", "msgpu::k_select_anchor_off",
              "msgpu::k_scatter_", "msgpu::k_rank_anchor", "msgpu::k_max_ids"),
    "candidates": ("msgpu::k_candidates", "msgpu::k_cand_reduce", "msgpu::k_emit_edges", "msgpu::k_classify_reads", "msgpu::k_bound"),
    "compact": ("msgpu::k_compact",),
}


# ---- CPU baseline ----------------------------------------------------------------------------------------------------------

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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, budget_reads, cores):
    """The CPU side of the comparison: the C oracle (a single-thread restatement of the reference's algorithm) on a
    bounded sample of the same workload shape -- once on one core, and once as `cores` independent samples (different
    seeds), one per core, at the same time.  The second figure is what a perfectly scaling multi-threaded CPU build
    could reach on those cores; the reference's own ThreadPool fan-out gets SLOWER with threads."""
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
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {
        "value": e1 / dt1, "unit": "overlap-pairs/s", "cores": 1, "kind": "port", "cpu_model": _cpu_model(),
        "kind_detail": "port (oracle/ms_oracle.c, a single-thread C restatement; the reference itself does not build here: it "
                       "needs microsoft/GSL, which is neither under /root/reference nor in the image)",
        "cores_available": avail,
        "core_policy": "min(16, cores this process may run on): 16 = the GPU box's CPU share per GPU",
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
            "one_core_quarter_sample_value": out["value"], "value": sum(r[0] for r in res) / busy, "cores": cores,
            "sample": "%d independent samples of that shape (seeds differ), one per core, run together: %d edges in "
                      "%.2f s (slowest worker; %.2f s with process start-up); one core alone: %s" % (
                          cores, sum(r[0] for r in res), busy, wall, out["sample"]),
        })
    return out


def cpu_one_core_full(workload):
    """The C oracle ONCE on the line's own workload (every row of it), one core, in a worker process of its own."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    w = WORKLOADS[workload]
    with ProcessPoolExecutor(max_workers=1, mp_context=mp.get_context("spawn")) as pool:
        e, m, c, dt = pool.submit(_cpu_sample, (w["n_reads"], w["read_len"], w["n_anchors"], w["seed"])).result()
    return {"value": e / dt, "unit": "overlap-pairs/s", "cores": 1, "kind": "port",
            "sample": "the workload of this line, whole: %d edges, %d EdgeMatches, %d compat checks in %.2f s on one core" % (e, m, c, dt)}


def cpu_consensus_baseline():
    """The consensus half on the CPU: everything behind the overlap tables -- contraction test, graph clean-up,
    getDirectedGraph, linearizeGraph, assemblePath incl. the string stitching -- through the oracles (C for the contraction
    test, the Python restatements oracle/ms_graph_py.py and ms_assemble_py.py for the rest: they are the only CPU statement
    of those stages that exists here), one core, on the tiled shape at the size of BASELINE.json configs[1]."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ms_oracle_ctypes as oracle
    from oracle import ms_graph_py as G
    from oracle.ms_assemble_py import assemble_path
    from muchsalsa_amd import synth
    shape = synth.TILED["cfg2"]
    rows, read_names, anchor_names = synth.accepted_rows(synth.paf_table(**shape))
    n_reads, L, seed = shape["n_reads"], shape["read_len"], shape["seed"]
    Gn, r_start, r_fwd = synth.read_layout(n_reads, L, seed, read_len_min=shape["read_len_min"])
    r_len = synth.read_lengths(n_reads, L, seed, shape["read_len_min"])
    a_start, a_len = synth.anchor_layout(n_reads, L, 0, seed, tiled=True)
    genome = synth.genome_bases(Gn, seed).tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    nano, illu = {}, {}
    for i, nm in enumerate(read_names):
        o = int(nm[1:])
        sq = genome[r_start[o]: r_start[o] + r_len[o]]
        nano[i] = sq if r_fwd[o] else sq.translate(comp)[::-1]
    for j, nm in enumerate(anchor_names):
        o = int(nm[1:])
        illu[j] = genome[a_start[o]: a_start[o] + a_len[o]]
    t = oracle.overlap(rows)
    t0 = time.perf_counter()
    co = oracle.find_contraction_edges(t, len(t["read_len"]))
    vm = {(int(r["read_id"]), int(r["anchor_id"])): r for r in rows}
    g, el, oo = G.build_graph(t, t["read_len"], t["read_first_line"])
    contain = G.clean_up(g, el, oo, co, lambda r, a: (r, a) in vm)
    edges, ems = t["edges"], t["ems"]
    eidx = {(int(e["v1"]), int(e["v2"])): i for i, e in enumerate(edges)}

    def em_of(a, b):
        e = edges[eidx[(a, b)]]
        return {int(m["anchor_id"]): (int(m["ov_lo"]), int(m["ov_hi"]))
                for m in ems[int(e["em_off"]): int(e["em_off"]) + int(e["em_cnt"])]}
    oc = {v: [dict(nano=c["nano"], dir=c["direction"], matches={a: vm[(c["nano"], a)] for a in c["anchors"]})
              for c in lst] for v, lst in contain.items()}
    res = [assemble_path(p, st, vm, oc, nano, illu, k) for k, (p, st) in enumerate(G.assemble_all(g, em_of))]
    dt = time.perf_counter() - t0
    T = sum(len(r["target"]) for r in res)
    return {"value": T / dt / 1e6, "unit": "consensus-Mbases/s", "cores": 1,
            "kind": "port (PYTHON restatements of src/main.cpp:194-310, dg.cpp, lg.cpp, ap.cpp; C for findContractionEdges): an "
                    "interpreter's speed, a weak bar -- the only figure for the reference's own code is the survey's 0.23 "
                    "consensus-Mbases/s, measured on another machine (see sample)",
            "sample": "tiled shape at configs[1] size (%d reads, %d unitigs, %d edges): %d contigs, %d bases in %.2f s; "
                      "the reference itself, measured by the survey on a 1 Mb genome: 0.23 consensus-Mbases/s (SURVEY.md "
                      "section 6)" % (n_reads, len(anchor_names), len(edges), len(res), T, dt)}


# ---- the legs reported beside the metric ---------------------------------------------------------------------------------

def host_to_host_leg(ctx, rows, n_batches, reps=6):
    """SURVEY section 8(d) / BASELINE.md 3.4: wall time from [row table in pinned host memory] to [edge, EdgeMatch, order
    and id tables in pinned host memory], through msgpu_overlap_batched (the ThreadPool replacement)."""
    from muchsalsa_amd import overlap
    pinned = overlap.PinnedRows(rows)
    ctx.set_stream(None)  # the context's own compute stream next to its copy stream (the timed steps ran on torch's)
    try:
        ctx.overlap_batched(pinned, n_batches, copy=False)  # warm-up: pinned result arena, second table set
        walls, infos = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            t, info = ctx.overlap_batched(pinned, n_batches, copy=False)
            walls.append(1e3 * (time.perf_counter() - t0))
            infos.append(info)
        k = int(np.argsort(walls)[len(walls) // 2])
        n_edges, nbytes = int(len(t["edges"])), int(sum(t[x].nbytes for x in ("edges", "ems", "orders", "ids")))
        ok = bool(np.array_equal(t["edges"]["em_off"], np.concatenate([[0], np.cumsum(t["edges"]["em_cnt"])[:-1]]))
                  and np.array_equal(t["ems"]["edge_idx"][:: max(1, len(t["ems"]) // 1000)],
                                     np.repeat(np.arange(n_edges, dtype=np.uint32), t["edges"]["em_cnt"])[
                                         :: max(1, len(t["ems"]) // 1000)]))
        wall = walls[k]
        h2d = rows.nbytes
        # SURVEY 8(d) literally: "row table in host memory -> order/edge tables in host memory".  The EdgeMatch table (5/6 of
        # the bytes) stays in HBM (MSGPU_BATCH_NO_EDGEMATCHES); downstream only assemblePath reads EdgeMatches, and only the
        # path edges' (msgpu_get_edgematches).  This is what muchsalsa_amd.pipeline.run / msgpu::assemble call.
        ctx.overlap_batched(pinned, 0, copy=False, resident=True, edgematches=False)  # warm-up: job tables in HBM
        lw, li = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            t2, info2 = ctx.overlap_batched(pinned, 0, copy=False, resident=True, edgematches=False)  # 0 = the library's choice
            lw.append(1e3 * (time.perf_counter() - t0))
            li.append(info2)
        k2 = int(np.argsort(lw)[len(lw) // 2])
        # ... and with the rows in the 28-byte link form (msgpu_pack_rows: read lengths once per read, lines as runs, flags in
        # the score's top bits; expanded in HBM by one kernel): what a loader that writes this form hands over
        packed_leg = None
        try:
            packed = overlap.PackedRows(rows, int(rows["read_id"].max()) + 1 if len(rows) else 0)
            try:
                ctx.overlap_batched(packed, 0, copy=False, resident=True, edgematches=False)
                pw, pi = [], []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    t3, info3 = ctx.overlap_batched(packed, 0, copy=False, resident=True, edgematches=False)
                    pw.append(1e3 * (time.perf_counter() - t0))
                    pi.append(info3)
                k3 = int(np.argsort(pw)[len(pw) // 2])
                packed_leg = {"ms": pw[k3], "overlap_pairs_per_s": int(len(t3["edges"])) / (pw[k3] * 1e-3), "load_ms": pi[k3]["load_ms"],
                              "rows_bytes_h2d": int(packed.link_bytes), "line_runs": int(packed.n_runs),
                              "ms_samples": [round(x, 3) for x in pw],
                              "tables_equal_unpacked_run": bool(all(t3[x].tobytes() == t2[x].tobytes() for x in ("edges", "orders", "ids"))),
                              "stage": "the same region with the row table in its 28-byte link form (msgpu_row28: %d bytes up instead "
                                       "of %d), msgpu_overlap_batched_ex(MSGPU_BATCH_ROWS_PACKED | MSGPU_BATCH_NO_EDGEMATCHES)" % (
                                           int(packed.link_bytes), int(rows.nbytes))}
            finally:
                packed.close()
        except Exception as exc:  # noqa: BLE001 -- (a table that does not pack: the 40-byte figures stand alone)
            packed_leg = {"error": "%s: %s" % (type(exc).__name__, exc)}
        lean_bytes = int(sum(t2[x].nbytes for x in ("edges", "orders", "ids")))
        lean_ok = bool(t2["ems"] is None and all(t2[x].tobytes() == t[x].tobytes() for x in ("edges", "orders", "ids")))
        pick = np.random.default_rng(1).integers(0, max(n_edges, 1), 10_000).astype("<u4") if n_edges else np.zeros(0, "<u4")
        t0 = time.perf_counter()
        off, sel = ctx.get_edgematches(pick, copy=False)
        t_sel = 1e3 * (time.perf_counter() - t0)
        if n_edges:
            probe = pick[:64]
            lean_ok = lean_ok and all(
                sel[int(off[i]): int(off[i + 1])].tobytes() == t["ems"][int(t["edges"]["em_off"][e]): int(t["edges"]["em_off"][e]) + int(t["edges"]["em_cnt"][e])].tobytes()
                for i, e in enumerate(probe))
        lean = {"ms": lw[k2], "overlap_pairs_per_s": n_edges / (lw[k2] * 1e-3), "table_bytes_d2h": lean_bytes,
                "load_ms": li[k2]["load_ms"], "compute_done_ms": li[k2]["compute_done_ms"], "batches": int(li[k2]["n_batches"]),
                "ms_samples": [round(x, 3) for x in lw], "tables_equal_full_run": lean_ok,
                "edgematches_left_in_hbm": int(li[k2]["n_ems"]),
                "get_edgematches_of_10k_edges_ms": t_sel, "edgematches_fetched": int(off[-1]) if len(off) else 0,
                "floor": "%.0f MB up + %.0f MB down at ~55 GB/s = %.1f ms, + index and first window" % (
                    h2d / 1e6, lean_bytes / 1e6, (h2d + lean_bytes) / 55e9 * 1e3),
                "stage": "rows in pinned host memory -> msgpu_overlap_batched_ex(MSGPU_BATCH_NO_EDGEMATCHES) -> edge, order "
                         "and id tables in pinned host memory, EdgeMatch table resident in HBM; what pipeline.run calls",
                "packed_rows": packed_leg}
        return {"ms": wall, "without_edgematches": lean, "overlap_pairs_per_s": n_edges / (wall * 1e-3), "batches": int(infos[k]["n_batches"]),
                "rows_bytes_h2d": int(h2d), "table_bytes_d2h": nbytes,
                "load_ms": infos[k]["load_ms"], "first_batch_ms": infos[k]["first_batch_ms"],
                "compute_done_ms": infos[k]["compute_done_ms"], "ms_samples": [round(x, 3) for x in walls],
                "link_gbs_over_whole_call": (h2d + nbytes) / (wall * 1e-3) / 1e9,
                "tables_consistent": ok,
                "floor": "the link: %.0f MB up + %.0f MB down at the ~55 GB/s PCIe Gen5 x16 delivers here = %.1f ms; the "
                         "index build and the first batch (~1 ms) cannot hide behind a copy" % (
                             h2d / 1e6, nbytes / 1e6, (h2d + nbytes) / 55e9 * 1e3),
                "stage": "rows in pinned host memory -> msgpu_overlap_batched (H2D once, index once, %d windows of owner "
                         "reads; window k's tables copied out while window k+1 computes) -> four tables in pinned host "
                         "memory; median of %d calls" % (int(infos[k]["n_batches"]), reps)}
    finally:
        pinned.close()


def group_leg(rows, reps=4):
    """msgpu_group_overlap with the devices this process may use as members (here: one): rows in pinned host memory -> every
    member's HBM -> index + shard -> wire-form slab -> ONE grouped RCCL all-gather -> merge -> merged list in host memory."""
    from muchsalsa_amd import overlap
    pinned = overlap.PinnedRows(rows)
    try:
        with overlap.OverlapGroup([0]) as grp:
            grp.overlap(pinned, copy=False)  # warm-up: communicator, buffers
            best = None
            for _ in range(reps):
                t, info = grp.overlap(pinned, copy=False)
                if best is None or info["wall_ms"] < best["wall_ms"]:
                    best = dict(info, n_edges=int(len(t["edges"])))
        return {"members": int(best["n_members"]), "wall_ms": best["wall_ms"], "compute_ms": best["compute_ms"],
                "exchange_ms": best["exchange_ms"], "overlap_pairs_per_s": best["n_edges"] / (best["wall_ms"] * 1e-3),
                "slab_bytes": int(best["slab_bytes"]), "id_bytes": int(best["id_bytes"]),
                "stage": "msgpu_group_overlap (C++, one process): rows in pinned host memory -> HBM of every member over its own link "
                         "-> index + shard v1 %% n -> msgpu_pack_wire -> ncclGroupStart / ncclAllGather per member / ncclGroupEnd -> "
                         "msgpu_merge_wire -> merged edge, order and id tables in host memory; %d member(s) here: a rehearsal of the "
                         "path, not a scaling measurement" % int(best["n_members"])}
    finally:
        pinned.close()


def group_on_node(world, workload, timeout_s=300):
    """msgpu_group_overlap over the node's `world` GPUs -- the in-process C++ driver of the N > 1 path (a libms caller's view:
    include/msgpu.h msgpu_group_*, RCCL resolved by the library) -- run as a CHILD process of rank 0 after the ranks' own process
    group is gone: a failure or a hang of a collective that has never run on hardware cannot take the line with it."""
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "group_rehearsal.py"), "--devices",
           ",".join(str(d) for d in range(world)), "--workload", workload, "--json"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MSGPU_GROUP_TRANSPORT")}
    try:
        time.sleep(1.0)  # (the other ranks' processes let go of their devices)
        run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        if run.returncode != 0 or not lines:
            return {"error": "exit code %d: %s" % (run.returncode, run.stderr[-600:])}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s (the child process was killed)" % timeout_s}
    except Exception as exc:  # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


def sharded_host_to_host_leg(torch, dist, D, dev, ctx, rows, world, rank, n_batches, reps=5):
    """SURVEY 8(d)'s region -- rows in host memory -> tables in host memory -- over the node's N PCIe links instead of one:
      rows      every rank holds the job's row table in its own pinned host memory, uploads 1/N of it over ITS link, and
                ONE all-gather over xGMI completes the table in every HBM (a second, input-side collective: xGMI moves the
                202 MB of configs[2] in well under the 3.7 ms one PCIe link needs for them);
      compute   msgpu_overlap_batched_ex on the rank's shard (edges with v1 % N == rank), windows on two streams, tables
                resident; the shard's four tables travel to the rank's pinned host memory over its own link;
      merge     edges | orders | ids in wire form through the one-collective SlabExchange + msgpu_merge_wire: the merged edge list
                in every HBM (what msgpu_find_contraction_edges takes); EdgeMatch tables stay rank-local.
    Wall = max over ranks; the result tables are distributed over the ranks' host memories."""
    from muchsalsa_amd import overlap
    from muchsalsa_amd._lib import EDGE_DTYPE, ORDER_DTYPE
    n = len(rows)
    per = (n + world - 1) // world
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    pin = torch.empty(per * rows.itemsize, dtype=torch.uint8).pin_memory()
    pin[: (hi - lo) * rows.itemsize] = torch.from_numpy(rows[lo:hi].view(np.uint8).copy())
    d_slice = torch.empty(per * rows.itemsize, dtype=torch.uint8, device=dev)
    d_rows = torch.empty(world * per * rows.itemsize, dtype=torch.uint8, device=dev)
    id_bytes = 3 if int(rows["anchor_id"].max(initial=0)) < 1 << 24 else 4  # (every rank holds the job's rows: same decision)
    exchange = D.SlabExchange(dev, wire=id_bytes)  # the wire form of include/msgpu.h
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    if world > 1:
        ctx.set_shard(rank, world)
    keep = {}

    def once():
        with torch.cuda.stream(stream):
            d_slice.copy_(pin, non_blocking=True)                   # 1/N of the rows over this rank's link
            dist.all_gather_into_tensor(d_rows, d_slice)            # ... the rest over xGMI
            t, info = ctx.overlap_batched(None, n_batches, copy=False, resident=True, device_rows=(d_rows.data_ptr(), n))
            c = ctx.counts()

            def fill(slab, offs):
                ctx.pack_wire(slab.data_ptr() + offs[0], slab.data_ptr() + offs[1], slab.data_ptr() + offs[2], id_bytes=id_bytes)
            gathered, allc, offs, slab_bytes = exchange.gather((c.n_edges, c.n_orders, c.n_ids), fill)
            tot = allc.sum(axis=0)
            need = (max(int(tot[0]), 1) * EDGE_DTYPE.itemsize, max(int(tot[1]), 1) * ORDER_DTYPE.itemsize, max(int(tot[2]), 1) * 4)
            if "bufs" not in keep or any(b.numel() < k for b, k in zip(keep["bufs"], need)):  # (the merged tables: kept from run to run)
                keep["bufs"] = tuple(torch.empty(int(k * 1.05) + 256, dtype=torch.uint8, device=dev) for k in need)
            m_e, m_o, m_i = keep["bufs"]
            ctx.merge_wire(gathered.data_ptr(), allc, slab_bytes, offs, m_e.data_ptr(), m_o.data_ptr(), m_i.data_ptr(),
                           id_bytes=id_bytes)
            stream.synchronize()
            keep.update(t=t, info=info, allc=allc, c=c)

    once()  # warm-up: arenas, capacity agreement
    once()
    walls = []
    for _ in range(reps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        once()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        walls.append(1e3 * float(dt.item()))
    ctx.set_stream(None)
    k = int(np.argsort(walls)[len(walls) // 2])
    t, allc = keep["t"], keep["allc"]
    n_edges = int(allc[:, 0].sum())
    mine = int(sum(t[x].nbytes for x in ("edges", "ems", "orders", "ids")))
    return {"ms": walls[k], "overlap_pairs_per_s": n_edges / (walls[k] * 1e-3), "ms_samples": [round(x, 3) for x in walls],
            "edges": n_edges, "rows_bytes_h2d_this_rank": int(per * rows.itemsize), "table_bytes_d2h_this_rank": mine,
            "collectives_per_call": 2, "windows": int(keep["info"]["n_batches"]),
            "scaling": "strong",
            "stage": "rows in every rank's pinned host memory -> 1/N H2D per rank + one xGMI all-gather of the row table -> "
                     "msgpu_overlap_batched_ex on the rank's shard, its four tables to the rank's pinned host memory -> one "
                     "all-gather + merge of edges | orders | ids; wall = max over ranks, median of %d calls" % reps}


def consensus_leg(torch, dev, world, rank, w, steps, warmup):
    """Device half of the consensus stage (A9) on the same synthetic reads: slice / reverse-complement / stitch.

    Sequences: a random genome; read i = genome[start_i : start_i + L], reverse-complemented for '-' reads (built on the
    device with the gather kernel itself and installed as the nanopore store).  Layout (host, numpy, not timed): reads in
    genome order stitched by the updateConsensusBase append rule (ap.cpp:205-229: a read contributes the part that
    extends the contig) -> target contigs; every read, oriented to the genome strand, is a query.  Timed: the gather
    kernel producing target+queries.  Checked at full size: the stitched target equals the genome, byte for byte.
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


def build_stores(torch, dev, w, read_names, anchor_names):
    """The two sequence stores of the workload (a synth.CONFIGS / synth.TILED shape), keyed by Registry id, cut out of the
    synthetic genome on the device and packed to 2 bits per base.  -> (store, read starts, read strands, genome length)
    by Registry id"""
    from muchsalsa_amd import sequences as S, synth
    from muchsalsa_amd._lib import COPY_DTYPE, COPY_ILLUMINA, COPY_REVCOMP
    n_reads, L, seed = w["n_reads"], w["read_len"], w["seed"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, seed, read_len_min=w.get("read_len_min"))
    r_len = synth.read_lengths(n_reads, L, seed, w.get("read_len_min"))
    a_start, a_len = synth.anchor_layout(n_reads, L, w["n_anchors"], seed, tiled=w.get("tiled", False))
    read_orig = np.array([int(n[1:]) for n in read_names])      # Registry id -> generator index
    anchor_orig = np.array([int(n[1:]) for n in anchor_names])
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    genome = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[
        torch.randint(0, 4, (G,), device=dev, generator=gen)]
    store = S.SeqStore(device=dev.index)
    store.upload_device(S.ILLUMINA, genome.data_ptr(), G, [0], [G])
    rs, rf, rl = r_start[read_orig], r_fwd[read_orig], r_len[read_orig].astype(np.uint64)
    roff = np.concatenate([[0], np.cumsum(rl)[:-1]]).astype(np.uint64)
    mk = np.zeros(len(read_orig), dtype=COPY_DTYPE)
    mk["src_off"], mk["dst_off"], mk["len"] = rs, roff, rl
    mk["flags"] = COPY_ILLUMINA | np.where(rf, 0, COPY_REVCOMP).astype(np.uint32)
    d_reads = torch.empty(int(rl.sum()), dtype=torch.uint8, device=dev)
    store.run(store.plan(mk), d_reads.data_ptr(), d_reads.numel())
    al = a_len[anchor_orig].astype(np.uint64)
    aoff = np.concatenate([[0], np.cumsum(al)[:-1]]).astype(np.uint64)
    mk = np.zeros(len(anchor_orig), dtype=COPY_DTYPE)
    mk["src_off"], mk["dst_off"], mk["len"], mk["flags"] = a_start[anchor_orig], aoff, al, COPY_ILLUMINA
    d_anch = torch.empty(int(al.sum()), dtype=torch.uint8, device=dev)
    store.run(store.plan(mk), d_anch.data_ptr(), d_anch.numel())
    store.synchronize()
    store.upload_device(S.NANOPORE, d_reads.data_ptr(), d_reads.numel(), roff, rl)
    store.upload_device(S.ILLUMINA, d_anch.data_ptr(), d_anch.numel(), aoff, al)
    del d_reads, d_anch
    store.pack()
    return store, rs, rf, G


def assemble_paths(store, rows, prepared, threads, band=64, reps=3):
    """assemblePath over prepared path inputs, timed: install the VertexMatch table (msgpu_assembly_borrow_rows, once per
    job) + host layout of every path (msgpu_assembly_add_paths) + ONE gather + FASTA wrapping with the texts copied back
    (msgpu_assembly_finish).  Medians of `reps` fresh assemblies.  Self-check: msgpu_assembly_validate (banded edit
    distance of every query against the stretch of its contig its PAF line names)."""
    from muchsalsa_amd.assembly import Assembly
    warm = Assembly(store)  # warm-up (untimed), like the W warm-up steps of the overlap half
    warm.set_rows(rows, copy=False)
    warm.add_prepared_batch(prepared, threads)
    warm.finish()
    warm.close()
    samples, asm = [], None
    for _ in range(reps):
        if asm is not None:
            asm.close()
        asm = Assembly(store)
        t0 = time.perf_counter()
        asm.set_rows(rows, copy=False)  # msgpu_assembly_borrow_rows: the PAF loader's table stays where it is
        t_i = time.perf_counter() - t0
        t0 = time.perf_counter()
        status = asm.add_prepared_batch(prepared, threads)
        t_l = time.perf_counter() - t0
        t0 = time.perf_counter()
        asm.finish()
        t_d = time.perf_counter() - t0
        samples.append((t_l, t_d, t_i))
    t_layout, t_device, t_index = (float(np.median([x[k] for x in samples])) for k in range(3))
    info, qinfo = asm.paths, asm.queries
    T, Q = int(info["target_len"].sum()), int(qinfo["len"].sum())
    asm.validate(band)  # warm-up
    t0 = time.perf_counter()
    dist, cells = asm.validate(band)
    t_val = time.perf_counter() - t0
    os.environ["MSGPU_ED_DP"] = "1"  # the banded anti-diagonal DP kernel on the same pairs: same numbers, its time beside
    try:
        asm.validate(band)
        t0 = time.perf_counter()
        dist_dp, _ = asm.validate(band)
        t_dp = time.perf_counter() - t0
    finally:
        del os.environ["MSGPU_ED_DP"]
    tot_ms = 1e3 * (t_index + t_layout + t_device)
    res = {"paths": int(len(info)), "paths_rejected": int((status != 0).sum()), "target_bases": T, "query_bases": Q,
           "queries": int(len(qinfo)), "pieces": int(len(asm.pieces)),
           "row_index_ms": 1e3 * t_index, "layout_ms": 1e3 * t_layout, "layout_threads": threads,
           "device_ms": 1e3 * t_device, "total_ms": tot_ms,
           "consensus_mbases_per_s": T / (tot_ms * 1e-3) / 1e6 if tot_ms > 0 else 0.0,
           "consensus_mbases_per_s_without_row_index": T / (1e3 * (t_layout + t_device) * 1e-3) / 1e6 if T else 0.0,
           "timing": "medians of %d fresh assemblies; total = row_index + layout + device" % reps,
           "text_bytes": len(asm.text(0)) + len(asm.text(1)) + len(asm.text(2)),
           "validate": {"kernel": "k_edit_distance (furthest-reaching points: the DP over (edits, diagonal))", "band": band,
                        "pairs": int(len(dist)), "ms_incl_copies": 1e3 * t_val,
                        "bases_compared_per_s": float(Q) / t_val if t_val > 0 else 0.0,
                        "banded_dp": {"kernel": "k_edit_distance_dp (anti-diagonals over the band, MSGPU_ED_DP=1)",
                                      "ms_incl_copies": 1e3 * t_dp, "dp_cells": cells,
                                      "dp_gcells_per_s": cells / t_dp / 1e9 if t_dp > 0 else 0.0,
                                      "same_distances": bool(np.array_equal(dist, dist_dp))},
                        "equivalent_banded_dp_gcells_per_s": cells / t_val / 1e9 if t_val > 0 else 0.0,
                        "queries_within_band": int((dist <= band).sum()),
                        "median_distance": float(np.median(dist)) if len(dist) else None}}
    asm.close()
    return res


def aux_legs(torch, dev, args, w, ctx, rows, read_names, anchor_names):
    """What follows the chaining fan-out in main() on the job's own tables, never inside `value`:
    findContractionEdges on the GPU (tables still resident), the host graph stage (src/main.cpp:183-310), and
    assemblePath (a) over the paths linearizeGraph yields and (b) over chains that tile the whole synthetic genome."""
    from muchsalsa_amd import synth
    from muchsalsa_amd.assembly import Assembly
    from muchsalsa_amd.graph import GraphStage
    torch.cuda.synchronize()
    ctx.find_contraction_edges()  # warm-up: arena allocation
    t0 = time.perf_counter()
    contraction = ctx.find_contraction_edges()
    t_contr = time.perf_counter() - t0
    tables = ctx.tables()
    read_len, read_first = ctx.reads()
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    graph_leg = {"find_contraction_edges_ms_incl_copy_back": 1e3 * t_contr,
                 "contraction_edges": int((contraction >= 0).sum()), "shadow_edges": int(tables["edges"]["shadow"].sum())}
    best = None
    for _ in range(3):  # host times on a shared box are noisy: best of three fresh graphs
        t0 = time.perf_counter()
        gs = GraphStage(tables, read_len, read_first)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        gs.clean_up(contraction, None)
        t_clean = time.perf_counter() - t0
        t0 = time.perf_counter()
        gs.linearize(threads)
        t_lin = time.perf_counter() - t0
        if best is None or t_build + t_clean + t_lin < sum(best[:3]):
            if best is not None:
                best[3].close()
            best = [t_build, t_clean, t_lin, gs]
        else:
            gs.close()
    t_build, t_clean, t_lin, gs = best
    st = gs.stats
    graph_leg.update({"graph_build_ms": 1e3 * t_build, "clean_up_ms": 1e3 * t_clean, "linearize_ms": 1e3 * t_lin,
                      "host_total_ms": 1e3 * (t_build + t_clean + t_lin), "linearize_threads": threads,
                      "vertices_after": int(st.n_vertices), "edges_after": int(st.n_edges),
                      "decycled_edges": int(st.n_decycled_edges), "components": int(st.n_components),
                      "paths": int(st.n_paths), "path_reads": int(st.n_path_reads),
                      "stage": "flat-CSR host graph stage (csrc/graph_stage.cpp) on the job's 1 M-edge graph; best of 3"})
    ctx.close()  # give the arena back before the sequence buffers
    store, rs, rf, G = build_stores(torch, dev, w, read_names, anchor_names)
    # (a) the reference's own flow: the paths linearizeGraph found
    prepared = [(gs.path_input(i), gs) for i in range(gs.path_count)]
    if prepared:
        ap = assemble_paths(store, rows, prepared, threads)
        graph_leg["assemble_path_over_these_paths"] = ap
        sys_ms = 1e3 * t_contr + graph_leg["host_total_ms"] + ap["total_ms"]
        graph_leg["system_consensus_mbases_per_s"] = ap["target_bases"] / (sys_ms * 1e-3) / 1e6
        graph_leg["system_ms"] = sys_ms
        graph_leg["system_note"] = ("the consensus half as the flow runs it: contig bases / (findContractionEdges + host "
                                    "graph stage + assemblePath over the paths linearizeGraph yields), overlap tables resident")
    gs.close()
    # (b) chains tiling the genome (a window of it with --assemble-window-mb): the consensus stage at the size of the job
    window = int(args.assemble_window_mb * 1e6) if args.assemble_window_mb > 0 else G
    t0 = time.perf_counter()
    paths = synth.chain_paths(tables, rs, rf, w["read_len"], min(window, G), max_reads=12)
    t_paths = time.perf_counter() - t0
    prepared = [Assembly.prepare(p, stp, None, None, i) for i, (p, stp) in enumerate(paths)]
    asm_leg = assemble_paths(store, rows, prepared, threads)
    asm_leg.update({"reads_on_paths": int(sum(len(p) for p, _ in paths)), "window_mb": min(window, G) / 1e6,
                    "genome_mb": G / 1e6, "path_builder_ms_untimed": 1e3 * t_paths,
                    "stage": "assemblePath over chains of <= 12 reads tiling the synthetic genome (synth.chain_paths over "
                             "the tables the timed steps produced): VertexMatch table install + host layout of every "
                             "path + one gather + FASTA wrapping + copy-back of target.fa / query.fa"})
    store.close()
    return asm_leg, graph_leg


def n50(lengths):
    ls = np.sort(np.asarray(lengths, dtype=np.int64))[::-1]
    if not len(ls):
        return 0
    return int(ls[np.searchsorted(np.cumsum(ls), ls.sum() / 2.0)])


def tiled_leg(torch, dev, args, workload, threads, seed_offset=0):
    """NOT a BASELINE configuration: the same number of reads on the shape the graph stage and assemblePath exist for --
    unitigs that TILE the genome (no two anchors overlap, synth.TILED) and reads of mixed length (short ones contained in
    long ones).  Every stage of the flow produces a number on it: the overlap step (with the all-pairs-compatible
    shortcut), findContractionEdges hits, the graph stage, the paths, assemblePath, contig N50 against the genome."""
    from muchsalsa_amd import overlap, synth
    from muchsalsa_amd.graph import GraphStage
    shape = dict(synth.TILED[workload])
    shape["seed"] += seed_offset  # (N > 1: rank r runs partition r)
    rows, read_names, anchor_names = synth.accepted_rows(synth.paf_table(**shape))
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).to(dev)
    ctx = overlap.OverlapContext(device=dev.index)
    ctx.set_id_space(len(read_names), len(anchor_names))
    work = torch.cuda.Stream(device=dev)
    ctx.set_stream(work.cuda_stream)

    def step():
        ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
    with torch.cuda.stream(work):
        ctx.set_stage_events(False)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms_step = 1e3 * (time.perf_counter() - t0) / args.steps
        ctx.set_stage_events(True)
        step()
        tm = ctx.timings()
    c = ctx.counts()
    ctx.set_stream(None)
    pinned = overlap.PinnedRows(rows)
    ctx.overlap_batched(pinned, 0, copy=False, resident=True, edgematches=False)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        tables, _ = ctx.overlap_batched(pinned, 0, copy=False, resident=True, edgematches=False)
        walls.append(1e3 * (time.perf_counter() - t0))
    ctx.find_contraction_edges()
    t0 = time.perf_counter()
    contraction = ctx.find_contraction_edges()
    t_contr = 1e3 * (time.perf_counter() - t0)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        gs = GraphStage(tables, tables["read_len"], tables["read_first_line"])
        gs.clean_up(contraction, None)
        gs.linearize(threads)
        t1 = time.perf_counter()
        gs.set_path_edgematches(*ctx.get_edgematches(gs.path_edges(), copy=False))
        t2 = time.perf_counter()
        if best is None or t1 - t0 < best[0]:
            if best is not None:
                best[2].close()
            best = (t1 - t0, t2 - t1, gs)
        else:
            gs.close()
    t_graph, t_ems, gs = 1e3 * best[0], 1e3 * best[1], best[2]
    st = gs.stats
    store, rs, rf, G = build_stores(torch, dev, shape, read_names, anchor_names)
    ap = assemble_paths(store, rows, [(gs.path_input(i), gs) for i in range(gs.path_count)], threads)
    from muchsalsa_amd.assembly import Assembly
    asm = Assembly(store)
    asm.set_rows(rows, copy=False)
    asm.add_prepared_batch([(gs.path_input(i), gs) for i in range(gs.path_count)], threads)
    contig_len = asm.paths["target_len"].astype(np.int64)
    n_shadow = int(tables["edges"]["shadow"].sum())  # (`tables` are views of the context's pinned memory: read before close)
    del tables
    asm.close()
    gs.close()
    store.close()
    pinned.close()
    ctx.close()
    sys_ms = t_contr + t_graph + t_ems + ap["total_ms"]
    return {"workload": "NOT a BASELINE configuration: %d reads of %d..%d bp over a %d Mb genome covered end to end by %d "
                        "non-overlapping unitigs of 500..1500 bp (synth.TILED[%s]) -- the shape unitigs and long reads have"
                        % (shape["n_reads"], shape["read_len_min"], shape["read_len"], G // 1_000_000, len(anchor_names), workload),
            "rows": int(len(rows)), "edges": int(c.n_edges), "edgematches": int(c.n_ems), "orders": int(c.n_orders),
            "overlap_ms_per_step": ms_step, "overlap_pairs_per_s": c.n_edges / (ms_step * 1e-3),
            "stage_ms": {"index": tm.index_ms, "candidates": tm.candidates_ms, "chain_total": tm.chain_ms, "compact": tm.compact_ms},
            "edges_proven_clean": int(c.n_edges_fastpath),
            "host_to_host_without_edgematches_ms": float(np.median(walls)),
            "find_contraction_edges_ms_incl_copy_back": t_contr, "contraction_edges": int((contraction >= 0).sum()),
            "shadow_edges": n_shadow,
            "graph_stage_ms": t_graph, "path_edgematches_ms": t_ems, "vertices_after": int(st.n_vertices),
            "edges_after": int(st.n_edges), "components": int(st.n_components), "paths": int(st.n_paths),
            "path_reads": int(st.n_path_reads), "assemble_path": ap,
            "contigs": int(len(contig_len)), "contig_n50": n50(contig_len), "longest_contig": int(contig_len.max()) if len(contig_len) else 0,
            "genome_bases": int(G), "genome_covered_frac": float(contig_len.sum()) / G,
            "system_ms": sys_ms, "system_consensus_mbases_per_s": float(contig_len.sum()) / (sys_ms * 1e-3) / 1e6,
            "system_note": "contig bases / (findContractionEdges + graph stage + path EdgeMatches + assemblePath), tables resident"}


def write_e2e_inputs(d, w, tab):
    """BASELINE-shaped inputs of the whole executable as files in directory d: contigs.paf (the table as PAF text + the
    line the reference never parses), nanopore.fa, unitigs.fa.  Names are the generator's indices (the Registry does not
    care what a name looks like)."""
    import pandas as pd
    from muchsalsa_amd import synth
    n = len(tab["qname_id"])
    pd.DataFrame({"q": tab["qname_id"], "ql": tab["qlen"], "qs": tab["qstart"], "qe": tab["qend"],
                  "s": np.where(tab["strand"], "+", "-"), "t": tab["tname_id"], "tl": tab["tlen"], "ts": tab["tstart"],
                  "te": tab["tend"], "nm": tab["nmatch"], "bl": tab["qend"] - tab["qstart"],
                  "mq": np.full(n, 60)}).to_csv(os.path.join(d, "contigs.paf"), sep="\t", header=False, index=False)
    with open(os.path.join(d, "contigs.paf"), "a") as f:
        f.write("0\t1\t0\t1\t+\t0\t1\t0\t1\t0\t1\t0\n")  # the line the reference never parses (BlastFileReader.cpp:76)
    n_reads, L, seed = w["n_reads"], w["read_len"], w["seed"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, seed)
    a_start, a_len = synth.anchor_layout(n_reads, L, w["n_anchors"], seed)
    genome = synth.genome_bases(G, seed).tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with open(os.path.join(d, "nanopore.fa"), "wb") as f:
        for i in range(n_reads):
            sq = genome[r_start[i]: r_start[i] + L]
            f.write(b">%d\n" % i + (sq if r_fwd[i] else sq.translate(comp)[::-1]) + b"\n")
    with open(os.path.join(d, "unitigs.fa"), "wb") as f:
        for j in range(len(a_start)):
            f.write(b">%d\n" % j + genome[a_start[j]: a_start[j] + a_len[j]] + b"\n")


def e2e_leg(args, w, tab, read_names_all, threads):
    """The whole executable, files in -> files out, at the size of the workload: the PAF text and the two FASTA files of
    the BASELINE workload are written to a temporary directory (untimed) and muchsalsa_amd.pipeline.run (= main() of the
    reference, src/main.cpp:130-322) turns them into temp_1.{target.fa, query.fa, align.paf}.  Stage seconds as the
    driver measures them."""
    import shutil
    import tempfile
    from muchsalsa_amd import pipeline
    t_gen = time.perf_counter()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 8e9 else None
    d = tempfile.mkdtemp(prefix="msgpu_e2e_", dir=base)
    try:
        write_e2e_inputs(d, w, tab)
        in_bytes = {k: os.path.getsize(os.path.join(d, k)) for k in ("contigs.paf", "nanopore.fa", "unitigs.fa")}
        t_gen = time.perf_counter() - t_gen
        out = os.path.join(d, "out")
        os.mkdir(out)
        timings = {}
        t0 = time.perf_counter()
        res = pipeline.run(os.path.join(d, "contigs.paf"), os.path.join(d, "unitigs.fa"), os.path.join(d, "nanopore.fa"), out,
                           threads=threads, timings=timings)
        wall = time.perf_counter() - t0
        out_bytes = {k: os.path.getsize(os.path.join(out, k)) for k in ("temp_1.target.fa", "temp_1.query.fa", "temp_1.align.paf")}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"wall_s": wall, "stage_s": {k: round(v, 5) for k, v in timings.items()},
            "overlap_pairs_per_s": res["edges"] / wall, "consensus_mbases_per_s": res["target_bases"] / wall / 1e6,
            "counts": res, "input_bytes": in_bytes, "output_bytes": out_bytes, "threads": threads,
            "input_generation_s_untimed": t_gen, "tmpdir": "tmpfs" if base else "disk",
            "stage": "muchsalsa_amd.pipeline.run: PAF text + unitig FASTA + read FASTA in -> temp_1.{target.fa, query.fa, "
                     "align.paf} out; both rates are over the WHOLE run (parse, overlap, graph, sequences, assemblePath, write, teardown); "
                     "sequences_parse_* (file -> page-locked ring -> HBM) and sequences_pack_* run on one thread per file beside "
                     "parse_paf, sequences_registry on another beside overlap + graph: they overlap the other stages"}


# ---- N > 1: start the ranks -------------------------------------------------------------------------------------------------

def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run BEFORE this
    process touches the GPU, relay rank 0's JSON line, exit with the children's status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    return proc.returncode if proc.returncode != 0 or line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample-reads", type=int, default=25_000,
                    help="reads in the CPU-baseline sample (0 disables the baseline leg)")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="host cores of the CPU-baseline leg (0 = min(16, cores this process may run on))")
    ap.add_argument("--no-consensus", action="store_true", help="skip the consensus (sequence gather) leg")
    ap.add_argument("--kernels-only", action="store_true",
                    help="only the timed steps (profiling runs): no host-to-host, consensus, graph or CPU leg")
    ap.add_argument("--assemble-window-mb", type=float, default=0.0,
                    help="assemblePath leg: chain the reads starting in the first this-many Mb of the genome "
                         "(0 = the whole genome, the default; negative = skip the assemblePath / graph legs)")
    ap.add_argument("--no-tiled", action="store_true", help="skip the tiled-unitig leg (not a BASELINE configuration)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files-in / files-out leg (pipeline.run on the workload)")
    ap.add_argument("--batches", type=int, default=8, help="windows of the host-to-host leg (msgpu_overlap_batched)")
    ap.add_argument("--scaling", default="auto", choices=("auto", "weak", "strong"),
                    help="N > 1: weak = every rank runs one partition of an N-partition job (each of the workload's shape; "
                         "disjoint read / anchor id ranges), the merged edge list all-gathered on a communication stream "
                         "beside the next step's compute; strong = the ONE job of the workload sharded over the ranks by "
                         "v1 %% N (BASELINE.json configs[3]).  auto = strong is `value` (the configuration BASELINE names), the "
                         "weak figure and the rank-sharded host-to-host figure are reported beside it")
    ap.add_argument("--exchange-format", default="wire", choices=("wire", "whole"),
                    help="what the N > 1 exchange sends: the wire form of include/msgpu.h (17-byte edges, 33-byte orders; "
                         "msgpu_pack_wire / msgpu_merge_wire) or whole records (msgpu_copy_tables_device / msgpu_merge_gathered)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend of the N > 1 path: nccl = RCCL over xGMI (what is measured); gloo = a "
                         "REHEARSAL of the N > 1 control flow on hardware that cannot run N RCCL ranks")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (RCCL refuses two ranks on one device: use with --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 exchange path (all-gather + merge) even at world size 1 (used by the GPU tests)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the rank processes through torch.distributed.run even for --gpus 1 (tests the relay)")
    args = ap.parse_args()
    if args.kernels_only:
        args.no_consensus, args.cpu_sample_reads, args.assemble_window_mb = True, 0, -1.0
        args.no_tiled = args.no_e2e = True

    if (args.gpus > 1 or args.self_launch) and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, [a for a in sys.argv[1:] if a != "--self-launch"]))

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
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # started by a launcher (torchrun sets TORCHELASTIC_RUN_ID) = the distributed path, even with one rank
    multi = world > 1 or args.force_dist or "TORCHELASTIC_RUN_ID" in os.environ
    rccl_ranks = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            # RCCL's own stream at high priority: the all-gather's few workgroups must not queue behind a grid-filling
            # compute kernel of the next step, or "beside the compute" turns into "after it" (MSGPU_EXCHANGE_PRIORITY=0:
            # default priority; at world 1, where nothing crosses a link, the two measure the same)
            kw = {}
            try:
                if os.environ.get("MSGPU_EXCHANGE_PRIORITY", "2") != "0":
                    kw["pg_options"] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:  # noqa: BLE001 -- (a build without the option: default priority)
                kw = {}
            try:
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, **kw)
            except TypeError:  # (signature without pg_options: raised before anything was set up)
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        rccl_ranks = dist.get_world_size()  # what RCCL saw, not what the command line asked for

    w = WORKLOADS[args.workload]
    weak = multi and args.scaling in ("auto", "weak")
    # weak scaling: rank r holds partition r of an N-partition job -- the workload's shape, its own seed (rank 0: the
    # workload itself), its own reads and anchors (ids from 0; the merge adds the partition's id bases).  Reads of
    # different partitions share no anchor, so no edge crosses a partition: what sharding a genome by chromosome gives.
    seed = w["seed"] + (rank if weak else 0)
    tab = synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], seed)
    rows, read_names, anchor_names = synth.accepted_rows(tab)
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).to(dev)  # the accepted-row table, resident in HBM
    torch.cuda.synchronize()

    ctx = overlap.OverlapContext(device=local_rank)
    # One real (non-null) stream carries the compute: libmsgpu's kernels and torch's copies are ordered by it.
    # (high priority: the exchange's kernels on the communication stream fill what the compute leaves, not the other way round --
    # 2.85 -> 2.77 ms per step at world 1; MSGPU_WORK_PRIORITY=0: default priority)
    work = torch.cuda.Stream(device=dev, priority=0 if os.environ.get("MSGPU_WORK_PRIORITY") == "0" else -1)
    ctx.set_stream(work.cuda_stream)
    ctx.set_id_space(len(read_names), len(anchor_names))  # Registry sizes, known to whoever parsed the PAF
    merged_keep = {}

    def compute():
        ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        return ctx.counts()

    # what the exchange sends: 0 = whole records, 4 / 3 = the wire form with 4- / 3-byte anchor ids (3 while every rank's
    # anchor id space fits 24 bits -- decided per leg from numbers every rank holds, so every rank decides alike)
    form = {"wire": 0 if args.exchange_format == "whole" else 4}

    def fill(slab, offs):
        if form["wire"]:  # one pack kernel (+ the id copy with 4-byte ids) behind the compaction, on the compute stream
            ctx.pack_wire(slab.data_ptr() + offs[0], slab.data_ptr() + offs[1], slab.data_ptr() + offs[2],
                          id_bytes=form["wire"])
        else:
            ctx.copy_tables_device(d_edges=slab.data_ptr() + offs[0], d_orders=slab.data_ptr() + offs[1],
                                   d_ids=slab.data_ptr() + offs[2])

    def merge_slabs(*a, **kw):
        if form["wire"]:
            return ctx.merge_wire(*a, id_bytes=form["wire"], **kw)
        return ctx.merge_gathered(*a, **kw)

    def form_name():
        return "wire form (17-byte edges, 33-byte orders, %d-byte ids)" % form["wire"] if form["wire"] else "whole records"

    def timed_region(step, finish=None):
        """W warm-up steps, then K timed steps between barrier + synchronize on both sides; chain-kernel events of the
        timed steps; three more steps with stage markers.  -> (seconds, last step's result, chain kernel ms, stage ms)"""
        with torch.cuda.stream(work):
            ctx.set_stage_events(False)
            for _ in range(args.warmup):
                step()
            if finish:
                finish()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.timings()  # (starts a fresh window of chain-kernel event pairs)
            for _ in range(args.steps):
                res = step()  # no synchronisation between steps: the next index build queues up behind the compaction
            if finish:
                res = finish() or res
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            dt = time.perf_counter() - t0
            k_ms = ctx.timings().chain_kernel_ms  # HIP events around the chain kernels of every timed step, averaged
            ctx.set_stage_events(True)
            for _ in range(3):
                step()
            if finish:
                finish()
            tm = ctx.timings()
        return dt, res, k_ms, tm

    def over_ranks(dt):
        """-> (max over ranks, {min, max, per_rank} ms per step)"""
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, mine)
        per_rank = [1e3 * float(x) / args.steps for x in allt.cpu()]
        return max(float(x) for x in allt.cpu()), {"min": min(per_rank), "max": max(per_rank),
                                                   "per_rank": [round(x, 4) for x in per_rank]}

    def merged_tables(keep, k):
        tot = keep["tot"][k]
        me = keep["e"][k].cpu().numpy()[: int(tot[0]) * EDGE_DTYPE.itemsize].view(EDGE_DTYPE)
        mo = keep["o"][k].cpu().numpy()[: int(tot[1]) * ORDER_DTYPE.itemsize].view(ORDER_DTYPE)
        mi = keep["i"][k].cpu().numpy()[: int(tot[2]) * 4].view("<u4")
        return me, mo, mi

    def merged_consistent(me, mo, mi):
        return bool(
            np.array_equal(me["order_off"], np.concatenate([[0], np.cumsum(me["order_cnt"])[:-1]]).astype(np.uint64))
            and np.array_equal(mo["edge_idx"], np.repeat(np.arange(len(me), dtype=np.uint32), me["order_cnt"]))
            and np.array_equal(mo["ids_off"], np.concatenate([[0], np.cumsum(mo["ids_cnt"])[:-1]]).astype(np.uint64))
            and int(mo["ids_cnt"].sum()) == len(mi) and bool(np.all(mo["base"] == me["v1"][mo["edge_idx"]])))

    partition_rows = weak  # (rank > 0 holds a partition's rows, not the job's)
    weak_error = None
    exchange_info = strong_leg = sharded_h2h = weak_leg = None
    merge_ok = rank_ms = None
    scaling = "weak"
    if not multi:
        dt, c, k_ms, tm = timed_region(compute)
        n_edges_total = int(c.n_edges)
    elif weak:
        try:
            # ---- weak scaling: N partitions, the merge of batch k over xGMI beside the compute of batch k + 1 ----------------
            sizes = torch.tensor([len(read_names), len(anchor_names)], dtype=torch.int64, device=dev)
            alls = torch.empty(world * 2, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(alls, sizes)  # set-up, untimed: the partitions' id-space sizes
            alls = alls.cpu().numpy().reshape(world, 2)
            id_base = np.concatenate([np.zeros((1, 2), dtype=np.int64), np.cumsum(alls, axis=0)[:-1]]).astype("<u4")
            merged_keep.update(e=[None, None], o=[None, None], i=[None, None], tot=[None, None])

            def merge(gathered, allc, offs, slab_bytes, k, stream):
                tot = allc.sum(axis=0)
                for key, n in (("e", int(tot[0]) * EDGE_DTYPE.itemsize), ("o", int(tot[1]) * ORDER_DTYPE.itemsize), ("i", int(tot[2]) * 4)):
                    if merged_keep[key][k] is None or merged_keep[key][k].numel() < n:  # (first batches only)
                        merged_keep[key][k] = torch.empty(int(n * 1.125) + 256, dtype=torch.uint8, device=dev)
                merge_slabs(gathered.data_ptr(), allc, slab_bytes, offs, merged_keep["e"][k].data_ptr(),
                            merged_keep["o"][k].data_ptr(), merged_keep["i"][k].data_ptr(), id_base=id_base,
                            stream=stream.cuda_stream)
                merged_keep["tot"][k] = tot
                merged_keep["last"], merged_keep["allc"] = k, allc

            if form["wire"]:
                form["wire"] = 3 if int(alls[:, 1].max()) <= 1 << 24 else 4
            pe = D.PipelinedExchange(dev, merge, wire=form["wire"])
            if os.environ.get("MSGPU_EXCHANGE_GATE", "1") != "0":  # the merge of step k beside the chain stage of step k + 1
                pe.gate_arm = lambda: ctx.chain_launches() + 1
                pe.gate_wait = lambda count: ctx.wait_chain_launch(count, 1000)

            def step():
                c = compute()
                pe.submit((c.n_edges, c.n_orders, c.n_ids), fill)  # slab filled on the compute stream; all-gather on its own
                pe.collect()                                        # the batch before: headers, merge behind its all-gather
                return c

            def finish():
                pe.drain()

            dt, c, k_ms, tm = timed_region(step, finish)
            pe.close()
            allc = merged_keep["allc"]
            n_edges_total = int(allc[:, 0].sum())
            dt, rank_ms = over_ranks(dt)
            exchange_info = {"collectives_per_step": pe.collectives / max(1, pe.calls), "slab_bytes": int(pe.slab_bytes),
                             "format": form_name(),
                             "whole_record_slab_bytes": int(D.HEADER + D.slab_layout(pe.cap)[1]),
                             "regrows": pe.regrows, "overlapped": "all-gather + merge of step k on a communication stream, issued "
                                                                  "by a communication thread, beside the compute of step k + 1"}
            if rank == 0:  # not timed: the merged edge list is a consistent table whose first partition is our own
                me, mo, mi = merged_tables(merged_keep, merged_keep["last"])
                own = ctx.tables()
                ne, no, ni = (len(own[x]) for x in ("edges", "orders", "ids"))
                merge_ok = merged_consistent(me, mo, mi) and me[:ne].tobytes() == own["edges"].tobytes() \
                    and mo[:no].tobytes() == own["orders"].tobytes() and mi[:ni].tobytes() == own["ids"].tobytes() \
                    and bool(np.all(np.diff(me["v1"].astype(np.int64)) >= 0))  # id bases ascend: (v1, v2)-sorted
                if world > 1:
                    merge_ok = merge_ok and int(me["v1"][ne]) >= int(id_base[1][0]) and int(mi[ni:].min()) >= int(id_base[1][1])
        except Exception as exc:  # noqa: BLE001 -- (deterministic failures hit every rank alike: the strong leg is the line anyway)
            weak_error = "%s: %s" % (type(exc).__name__, exc)
            weak = False
            try:
                pe.close()  # (its communication thread must not go on polling the context beside the later legs)
            except Exception:  # noqa: BLE001
                pass
    if multi and (not weak or args.scaling == "auto"):
        # ---- strong scaling (BASELINE.json configs[3]): the ONE job of the workload, edges owned by v1 % N ---------------
        if partition_rows:  # every rank needs the job's table now (rank 0 holds it already)
            rows_s, rn_s, an_s = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"])) \
                if rank else (rows, read_names, anchor_names)
            d_rows_s = torch.from_numpy(rows_s.view(np.uint8).copy()).to(dev) if rank else d_rows
        else:
            rows_s, rn_s, an_s, d_rows_s = rows, read_names, anchor_names, d_rows
        ctx.set_id_space(len(rn_s), len(an_s))
        if world > 1:
            ctx.set_shard(rank, world)
        if form["wire"]:
            form["wire"] = 3 if len(an_s) <= 1 << 24 else 4
        exchange = D.SlabExchange(dev, wire=form["wire"])
        s_keep = {}

        def strong_step():
            ctx.load_rows_device(d_rows_s.data_ptr(), len(rows_s), keep_alive=d_rows_s)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            c = ctx.counts()
            # merge the edge list: ONE all-gather of the per-rank (header | edges | orders | ids) slab over xGMI, then the
            # HIP compaction / re-base kernel (msgpu_merge_gathered); everything on the compute stream
            gathered, allc, offs, slab_bytes = exchange.gather((c.n_edges, c.n_orders, c.n_ids), fill)
            tot = allc.sum(axis=0)
            need = (max(int(tot[0]), 1) * EDGE_DTYPE.itemsize, max(int(tot[1]), 1) * ORDER_DTYPE.itemsize, max(int(tot[2]), 1) * 4)
            if "bufs" not in s_keep or any(b.numel() < k for b, k in zip(s_keep["bufs"], need)):  # (the merged tables: kept from step to step)
                s_keep["bufs"] = tuple(torch.empty(int(k * 1.05) + 256, dtype=torch.uint8, device=dev) for k in need)
            m_e, m_o, m_i = s_keep["bufs"]
            merge_slabs(gathered.data_ptr(), allc, slab_bytes, offs, m_e.data_ptr(), m_o.data_ptr(), m_i.data_ptr())
            s_keep.update(e=[m_e], o=[m_o], i=[m_i], tot=[tot], allc=allc)
            return c

        s_dt, s_c, s_k_ms, s_tm = timed_region(strong_step)
        s_dt, s_rank_ms = over_ranks(s_dt)
        s_edges = int(s_keep["allc"][:, 0].sum())
        s_ok = None
        if rank == 0:
            me, mo, mi = merged_tables(s_keep, 0)
            s_ok = merged_consistent(me, mo, mi)
            if world == 1:
                own = ctx.tables()
                s_ok = s_ok and me.tobytes() == own["edges"].tobytes() and mo.tobytes() == own["orders"].tobytes() \
                    and mi.tobytes() == own["ids"].tobytes()
        strong_leg = {"scaling": "strong", "value": s_edges / (s_dt / args.steps), "unit": "overlap-pairs/s",
                      "ms_per_step": 1e3 * s_dt / args.steps, "edges": s_edges, "rank_ms_per_step": s_rank_ms,
                      "merged_edge_list_consistent": s_ok,
                      "exchange": {"collectives_per_step": exchange.collectives / max(1, exchange.calls),
                                   "slab_bytes": int(exchange.slab_bytes), "regrows": exchange.regrows,
                                   "format": form_name()},
                      "stage_ms": {"index": s_tm.index_ms, "candidates": s_tm.candidates_ms, "chain_total": s_tm.chain_ms,
                                   "chain_kernel": s_k_ms, "compact": s_tm.compact_ms},
                      "workload": "%s as ONE job: every rank indexes the whole row table (replicated), owns the edges with "
                                  "v1 %% %d == rank, one all-gather + merge per step on the compute stream "
                                  "(BASELINE.json configs[3] at N = 8)" % (w["name"], world)}
        # `value` at N > 1 is the STRONG figure -- BASELINE.json configs[3] is ONE 100k x 10 kb job sharded over the GPUs of a
        # node -- and the weak figure (N partitions of that shape: not a configuration BASELINE names) stands beside it
        if weak:
            weak_leg = {"scaling": "weak", "value": n_edges_total / (dt / args.steps), "unit": "overlap-pairs/s",
                        "ms_per_step": 1e3 * dt / args.steps, "edges": n_edges_total, "rank_ms_per_step": rank_ms,
                        "merged_edge_list_consistent": merge_ok, "exchange": exchange_info,
                        "stage_ms": {"index": tm.index_ms, "candidates": tm.candidates_ms, "chain_total": tm.chain_ms,
                                     "chain_kernel": float(k_ms), "compact": tm.compact_ms},
                        "workload": "%d partitions, each: %s (seeds %d..%d; disjoint read / anchor id ranges, so no edge crosses "
                                    "a partition), one partition per GPU; the all-gather + merge of step k on a communication "
                                    "stream beside the compute of step k + 1.  NOT a BASELINE configuration"
                                    % (world, w["name"], w["seed"], w["seed"] + world - 1)}
            weak = False
        scaling = "strong"
        dt, c, k_ms, tm, rank_ms, n_edges_total, merge_ok = s_dt, s_c, s_k_ms, s_tm, s_rank_ms, s_edges, s_ok
        exchange_info, strong_leg = strong_leg["exchange"], None
        try:
            sharded_h2h = sharded_host_to_host_leg(torch, dist, D, dev, ctx, rows_s, world, rank, args.batches)
        except Exception as exc:  # noqa: BLE001 -- (raised on every rank alike: same code, same sizes)
            sharded_h2h = {"error": "%s: %s" % (type(exc).__name__, exc)}
        ctx.set_shard(0, 1)
        ctx.set_id_space(len(read_names), len(anchor_names))
        if partition_rows:
            del d_rows_s

    h2h = asm_leg = graph_leg = grp_leg = None
    errors = {}
    if world == 1 and rank == 0 and not args.kernels_only:
        try:
            if args.batches >= 0 and not (args.batches == 0 and args.assemble_window_mb < 0):  # counter passes skip it
                h2h = host_to_host_leg(ctx, rows, args.batches)
        except Exception as exc:  # noqa: BLE001 -- the legs reported BESIDE the metric must never cost the metric line
            errors["host_to_host"] = "%s: %s" % (type(exc).__name__, exc)
        if h2h is not None:
            try:  # the in-process group (msgpu_group: what a C++ caller uses on a node) with one member: every step of its path
                grp_leg = group_leg(rows)
            except Exception as exc:  # noqa: BLE001
                errors["group"] = "%s: %s" % (type(exc).__name__, exc)
        if args.assemble_window_mb >= 0 and not args.no_consensus:
            try:
                ctx.load_rows_device(d_rows.data_ptr(), len(rows), keep_alive=d_rows)  # the job's tables again
                ctx.calculate_edges()
                ctx.chaining_and_overlaps()
                asm_leg, graph_leg = aux_legs(torch, dev, args, w, ctx, rows, read_names, anchor_names)
            except Exception as exc:  # noqa: BLE001
                errors["assemble_path"] = "%s: %s" % (type(exc).__name__, exc)
    cons = None
    if not args.no_consensus:
        ctx.close()  # give the arena back before the ~3 GB of sequence buffers
        try:
            cons = consensus_leg(torch, dev, world, rank, w, args.steps, args.warmup)
        except Exception as exc:  # noqa: BLE001
            if multi:
                raise  # the other ranks are waiting in the all-reduce below
            errors["consensus"] = "%s: %s" % (type(exc).__name__, exc)
    if cons is not None and multi:
        tt = torch.tensor([cons["ms"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cons["ms"] = float(tt.item())
    cons_weak = None
    if multi and not args.kernels_only and args.workload in synth.TILED:
        # the consensus half at N GPUs: partitions are independent (no collective on the path) -- every rank runs the flow
        # behind the overlap tables (contraction test, graph stage, path EdgeMatches, assemblePath) on its own partition of
        # the tiled shape; bases of all ranks / the slowest rank's time
        try:
            ctx.close()
            host_threads = max(1, min(16, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)) // max(1, world)))
            tl = tiled_leg(torch, dev, args, args.workload, host_threads, seed_offset=rank)
            mine = torch.tensor([tl["system_ms"], float(tl["assemble_path"]["target_bases"]), tl["overlap_ms_per_step"],
                                 float(tl["edges"])], dtype=torch.float64, device=dev)
        except Exception as exc:  # noqa: BLE001
            errors["consensus_weak"] = "%s: %s" % (type(exc).__name__, exc)
            mine = torch.tensor([float("nan")] * 4, dtype=torch.float64, device=dev)
        allm = torch.empty(world * 4, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allm, mine)
        allm = allm.cpu().numpy().reshape(world, 4)
        if not np.isnan(allm).any():
            cons_weak = {"scaling": "weak", "system_consensus_mbases_per_s": float(allm[:, 1].sum() / (allm[:, 0].max() * 1e-3) / 1e6),
                         "system_ms_per_rank": [round(float(x), 2) for x in allm[:, 0]], "contig_bases": int(allm[:, 1].sum()),
                         "overlap_pairs_per_s": float(allm[:, 3].sum() / (allm[:, 2].max() * 1e-3)),
                         "host_threads_per_rank": host_threads,
                         "stage": "every rank: findContractionEdges + host graph stage + path EdgeMatches + assemblePath on its own "
                                  "partition of the tiled shape (synth.TILED, seed + rank; NOT a BASELINE configuration); no "
                                  "collective on the path; contig bases of all ranks / the slowest rank's time"}
    tiled = e2e = None
    if world == 1 and rank == 0 and not args.kernels_only:
        host_threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        ctx.close()
        if not args.no_tiled and args.workload in synth.TILED:
            try:
                tiled = tiled_leg(torch, dev, args, args.workload, host_threads)
            except Exception as exc:  # noqa: BLE001
                errors["tiled_unitigs"] = "%s: %s" % (type(exc).__name__, exc)
        if not args.no_e2e:
            try:
                e2e = e2e_leg(args, w, tab, read_names, host_threads)
            except Exception as exc:  # noqa: BLE001
                errors["e2e"] = "%s: %s" % (type(exc).__name__, exc)
    del tab

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        # algorithmic bytes of the dominant kernels: 96 B per EdgeMatch (32 B EdgeMatch written +
        # two 32 B VertexMatch rows read) + 64 B per order + 4 B per id; one launch set processes all edges of the rank
        alg_bytes = 96 * c.n_ems + 64 * c.n_orders + 4 * c.n_ids
        k_ms = float(k_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src, valu = pmc_traffic(args.workload, world, CHAIN_KERNELS)
        if traffic is None and "lacks FETCH_SIZE" in str(traffic_src):  # (a capture of the build that still made a launch per width)
            traffic, traffic_src, valu = pmc_traffic(args.workload, world, CHAIN_KERNELS_BEFORE)
        roof = {"bound": "hbm", "kernel": "k_chain + k_chain_sub_all (one pass over the edges: a launch for the edges of 33..64 "
                                          "EdgeMatches, a launch for the three sub-wavefront widths)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": k_ms,
                "note": "kernel_ms = the two launches together (HIP events around them on the launch stream); these "
                        "kernels are bound by vector-ALU issue, not by HBM (valu_issue_frac)"}
        roof["time_share_of_value"] = k_ms / ms_per_step if ms_per_step > 0 else None
        if valu and k_ms > 0:
            # SQ_INSTS_VALU counts wave-instructions; one costs 4 cycles of its SIMD's issue (MI355X_MICROARCH.md); 1024 SIMDs
            insts = sum(valu.values())
            roof["valu_issue_frac"] = insts * 4.0 / (k_ms * 1e-3 * 2.4e9 * 1024)
            roof["valu_note"] = "SQ_INSTS_VALU of the four kernels (%s) x 4 cycles / (kernel_ms x 2.4 GHz x 1024 SIMDs)" % traffic_src
        # every stage against the HBM roofline: algorithmic bytes (DESIGN.md section 4's formulas) / HIP-event time of the stage
        # (stage markers of an extra step), counter traffic of the stage's kernels from the newest capture under profiles/
        R_rows, P_emit = int(len(rows)), int(c.n_ems)
        stage_alg = {"index": 40 * R_rows + 80 * R_rows,                       # rows in; by_read + by_anchor + scan view out
                     "candidates": 16 * R_rows + 32 * P_emit + 8 * P_emit,     # scan view + scaffold rows read, (j, t) written
                     "compact": 2 * (64 * int(c.n_orders) + 4 * int(c.n_ids))}  # order slots + ids read, dense tables written
        stage_t = {"index": tm.index_ms, "candidates": tm.candidates_ms, "compact": tm.compact_ms}
        roof_stages = {}
        for name, nbytes in stage_alg.items():
            t_ms = float(stage_t[name])
            tr, src, used = pmc_stage(args.workload, world, STAGE_KERNELS[name])
            ach = nbytes / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
            roof_stages[name] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "algorithmic_bytes": int(nbytes), "stage_ms": t_ms, "traffic": tr, "traffic_source": src,
                                 "traffic_kernels": used, "time_share_of_value": t_ms / ms_per_step if ms_per_step > 0 else None}
        roof_stages["index"]["path"] = {0: "bin (no global atomic per row)", 1: "atomic", 2: "two-pass"}.get(int(c.index_path) & 3, "?") + (
            " + generic scaffolds" if int(c.index_path) & 4 else "")
        out = {
            "metric": "overlap-pairs/s", "value": n_edges_total / (dt / args.steps), "unit": "overlap-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "int32+f64",
            "data": "synthetic",
            "config": {"workload": (w["name"] if world == 1 else
                                    "%d partitions, each: %s (seeds %d..%d; disjoint read / anchor id ranges, so no edge "
                                    "crosses a partition), one partition per GPU" % (world, w["name"], w["seed"], w["seed"] + world - 1)
                                    if weak else
                                    "%s -- ONE job sharded across %d GPUs: every rank indexes the replicated row table and owns the "
                                    "edges with v1 %% %d == rank, one RCCL all-gather + merge of the edge list per step "
                                    "(BASELINE.json configs[3] is this at N = 8)" % (w["name"], world, world)),
                       "rows": int(len(rows)), "reads": int(c.n_reads),
                       "anchors": int(c.n_anchors), "edges": n_edges_total, "edgematches_rank0": int(c.n_ems),
                       "orders_rank0": int(c.n_orders),
                       "parallelism": ("one job on one GPU" if not multi else
                                       "dp%d: one partition per rank, one RCCL all-gather + merge of the edge list per step on "
                                       "a communication stream" % world if weak else "edges sharded by v1 %% %d" % world),
                       "edges_proven_clean_rank0": int(c.n_edges_fastpath), "merged_edge_list_consistent": merge_ok,
                       "note": "value = the overlap half of the metric with the row table resident in HBM when the timed region "
                               "starts (the bench contract); survey_8d_region = SURVEY 8(d)'s own region, host memory to host "
                               "memory; host_to_host = that region's details; the consensus half is under 'consensus' "
                               "(gather kernel at full size), 'assemble_path' (assemblePath over the whole genome) and "
                               "'graph_stage' (the paths the real graph stage yields)"},
            "stage_ms": {"index": tm.index_ms, "candidates": tm.candidates_ms, "chain_total": tm.chain_ms,
                         "chain_kernel": k_ms, "compact": tm.compact_ms,
                         "note": "index / candidates / chain_total / compact: stage markers (HIP events) of an extra step "
                                 "after the timed ones; chain_kernel: mean over the timed steps"},
            "roofline": roof,
            "roofline_stages": roof_stages,
        }
        if multi:
            out["rccl_ranks"] = rccl_ranks if args.backend == "nccl" else None
            if args.backend != "nccl":
                out["rehearsal"] = "backend %s%s: control flow of the N > 1 path only, NOT a measurement" % (
                    args.backend, ", every rank on GPU 0" if args.single_device else "")
            out["rank_ms_per_step"] = rank_ms
            out["exchange"] = exchange_info
            if weak_error is not None:
                out["weak_scaling_error"] = weak_error + " -- the weak figure beside `value` is missing"
            if strong_leg is not None:
                out["strong"] = strong_leg
            if weak_leg is not None:
                out["weak"] = weak_leg
            if sharded_h2h is not None:
                out["host_to_host_sharded"] = sharded_h2h
            if cons_weak is not None:
                out["consensus_system"] = cons_weak
        if h2h is not None:
            out["host_to_host"] = h2h
            lean = h2h.get("without_edgematches") or {}
            # SURVEY.md section 8(d) / BASELINE.md 3.4 define overlap-pairs/s over [row table in host memory -> order / edge
            # tables in host memory].  The bench contract fixes `value` on inputs resident in HBM (the PCIe-inclusive rate "is
            # never value"), so that region's figures stand here, at the top level, beside it.
            # ... and as plain scalars inside `config` and `roofline`, the two objects the driver's record keeps whole
            # (BENCH_r04.json retained `survey_8d_region` as a bare key name only)
            pk0 = lean.get("packed_rows") or {}
            link_floor_ms = ((pk0.get("rows_bytes_h2d") or h2h.get("rows_bytes_h2d", 0)) + (lean.get("table_bytes_d2h") or 0)) / 55e9 * 1e3
            pk = lean.get("packed_rows") or {}
            best = pk if pk.get("ms") and pk.get("tables_equal_unpacked_run") and pk["ms"] < (lean.get("ms") or 1e30) else lean
            for holder in (out["config"], out["roofline"]):
                holder["survey_8d_ms"] = best.get("ms")
                holder["survey_8d_overlap_pairs_per_s"] = best.get("overlap_pairs_per_s")
                holder["survey_8d_rows_form"] = "28-byte link form (msgpu_row28)" if best is pk else "40-byte msgpu_row"
                holder["survey_8d_40_byte_rows_ms"] = lean.get("ms")
                holder["survey_8d_all_tables_ms"] = h2h.get("ms")
                holder["survey_8d_all_tables_overlap_pairs_per_s"] = h2h.get("overlap_pairs_per_s")
                holder["survey_8d_link_floor_ms"] = link_floor_ms
                holder["survey_8d_rows_bytes_h2d"] = (best.get("rows_bytes_h2d") if best is pk else h2h.get("rows_bytes_h2d"))
            out["config"]["survey_8d_note"] = (
                "SURVEY 8(d)'s own region: rows in pinned host memory -> edge / order / id tables in pinned host memory (EdgeMatch "
                "table left in HBM); all_tables = with the EdgeMatch table copied out too; link_floor = bytes up + down of the lean "
                "region at the ~55 GB/s this PCIe link delivers.  PCIe-inclusive, so never `value`")
            out["survey_8d_region"] = {
                "overlap_pairs_per_s": best.get("overlap_pairs_per_s"), "ms": best.get("ms"),
                "rows_form": out["config"]["survey_8d_rows_form"], "ms_with_40_byte_rows": lean.get("ms"),
                "region": "rows in pinned host memory -> edge, order and id tables in pinned host memory "
                          "(msgpu_overlap_batched_ex, MSGPU_BATCH_NO_EDGEMATCHES: the EdgeMatch table stays in HBM, fetched per "
                          "edge list by msgpu_get_edgematches); what pipeline.run / msgpu::assemble call",
                "all_four_tables_overlap_pairs_per_s": h2h.get("overlap_pairs_per_s"), "all_four_tables_ms": h2h.get("ms"),
                "note": "PCIe-inclusive: bounded by the host link (floor in host_to_host.floor), not by the kernels"}
        if grp_leg is not None:
            out["group"] = grp_leg
        if cons is not None:
            # algorithmic bytes on the 2-bit store: 0.25 B read + 1 B written per base (SURVEY 8(d) counted 1 B + 1 B for a
            # byte-per-base source; that figure is kept as "bytes_if_byte_store" for comparison)
            gb = 1.25 * (cons["target_bases"] + cons["query_bases"]) / 1e9
            g_gbs = gb / (cons["ms"] * 1e-3)
            g_traffic, g_src, _ = pmc_traffic(args.workload, world, ("msgpu::k_gather_packed",))
            out["consensus"] = {
                "stage": "slice / reverse-complement / stitch kernel k_gather_packed on the 2-bit sequence store "
                         "(layout precomputed on the host, not timed)",
                "gather_kernel_only_consensus_mbases_per_s": cons["target_bases"] / (cons["ms"] * 1e-3) / 1e6,
                "gather_kernel_only_query_mbases_per_s": cons["query_bases"] / (cons["ms"] * 1e-3) / 1e6,
                "note": "ONE kernel of the consensus half, the layout precomputed and untimed: not a consensus rate of the "
                        "system -- that is graph_stage.system_consensus_mbases_per_s (and e2e.consensus_mbases_per_s)",
                "target_bases": cons["target_bases"], "query_bases": cons["query_bases"], "pieces": cons["pieces"],
                "ms": cons["ms"], "verified_against_genome": cons["verified"],
                "roofline": {"bound": "hbm", "kernel": "k_gather_packed", "achieved": g_gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": g_gbs / HBM_PEAK_GBS, "traffic": g_traffic,
                             "traffic_source": g_src, "algorithmic_bytes_per_launch": int(gb * 1e9),
                             "bytes_if_byte_store": int(2 * (cons["target_bases"] + cons["query_bases"])),
                             "note": "2 bits in + 1 byte out per base; the same launch on a byte-per-base source would "
                                     "move 2 B per base"},
            }
        if graph_leg is not None:
            out["graph_stage"] = graph_leg
        if tiled is not None:
            out["tiled_unitigs"] = tiled
        if e2e is not None:
            out["e2e"] = e2e
        if asm_leg is not None:
            out["assemble_path"] = asm_leg
        for k, v in errors.items():
            out.setdefault(k, {})
            out[k] = {"error": v}
        if world == 1 and args.cpu_sample_reads > 0:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = args.cpu_cores if args.cpu_cores > 0 else max(1, min(16, avail))
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_reads, cores)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:  # one core on the workload of the line itself, beside the scaled samples
                full = cpu_one_core_full(args.workload)
                out["cpu_baseline"]["one_core_value"] = full["value"]
                out["cpu_baseline"]["one_core_sample"] = full["sample"]
                out["cpu_baseline"]["which_is_which"] = (
                    "value: %d scaled samples run together, one per core (what a perfectly scaling multi-threaded port would reach); "
                    "one_core_value: the C oracle on the WHOLE workload of this line, one core; one_core_quarter_sample_value: one "
                    "scaled sample alone.  kind 'port': the oracle is roughly 9x faster than the reference it restates -- a "
                    "CROSS-MACHINE ratio (the reference: 3.26 k overlap-pairs/s on the survey's 2.1 GHz Xeon container, SURVEY.md "
                    "section 6; the oracle: this box's host CPU), not a same-machine calibration" % cores)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"]["one_core_error"] = "%s: %s" % (type(exc).__name__, exc)
            try:
                out["cpu_baseline"]["consensus"] = cpu_consensus_baseline()
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"]["consensus"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    # The metric line must never be lost to what follows the measurement.  Ranks other than 0 tear down and leave.  Rank 0 arms
    # a watchdog that writes the line as it stands (and ends the process) should anything below stall, runs the group child under
    # its own, shorter limit, WRITES THE LINE, and only then tears down its context and process group: a hang or an abort in
    # RCCL's teardown comes after the line.
    if rank != 0:
        ctx.close()
        if multi:
            dist.destroy_process_group()
        return
    import threading
    written = threading.Lock()

    def write_line(extra=None):
        if not written.acquire(blocking=False):
            return
        if extra:
            out.update(extra)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    want_group = multi and args.backend == "nccl" and not args.single_device and not args.kernels_only
    group_limit_s = float(os.environ.get("MSGPU_BENCH_GROUP_LIMIT_S", "150"))

    def watchdog():
        write_line({"group_on_node": {"error": "rank 0 stalled behind the measurement (teardown or the group child): "
                                               "line written by the watchdog"}} if want_group else None)
        os._exit(0)

    guard = threading.Timer((group_limit_s if want_group else 0.0) + 60.0, watchdog)
    guard.daemon = True
    guard.start()
    try:
        ctx.close()  # our own arena: the group child gets GPU 0's memory
        if want_group:
            # the other ranks are leaving (or gone): the node's GPUs also go to ONE process, the C++ group, a child of this one
            out["group_on_node"] = group_on_node(world, args.workload, timeout_s=group_limit_s)
    except Exception as exc:  # noqa: BLE001
        out.setdefault("group_on_node", {"error": "%s: %s" % (type(exc).__name__, exc)})
    write_line()
    guard.cancel()
    if multi:
        try:
            dist.destroy_process_group()
        except Exception as exc:  # noqa: BLE001 -- (after the line: nothing to lose)
            print("destroy_process_group: %s" % exc, file=sys.stderr)


if __name__ == "__main__":
    main()
