"""The `muchsalsa` executable's main() (src/main.cpp:130-322) over the C-ABI: PAF + unitigs + long reads ->
temp_1.target.fa / temp_1.query.fa / temp_1.align.paf.

    stage (reference, src/main.cpp)                     here
    :153-157 BlastFileReader::read + calculateEdges      parse_paf (host) + OverlapContext.overlap_batched (HIP: the
    :170-178 chainingAndOverlaps fan-out                   ThreadPool replacement, EdgeMatch table left in HBM)
    :161-163 SequenceAccessor::buildIndex                SeqFile (host parse) + SeqStore.upload (HBM)
    :183-190 findContractionEdges                        OverlapContext.find_contraction_edges (HIP)
    :194-288 contraction, deletions, bitweight, MST, decycle   GraphStage.clean_up (host)
    :300-310 components -> getDirectedGraph -> linearizeGraph   GraphStage.linearize (host)
    :663-677 assemblePath per path                       Assembly.add_prepared_batch (host layout, threads)
             OutputWriter                                Assembly.finish (one gather + FASTA wrapping, HIP) + 3 writes
"""
import os
import threading
import time

import numpy as np

from . import overlap
from .assembly import Assembly
from .graph import GraphStage
from .sequences import ILLUMINA, NANOPORE, SeqFile, SeqStore


def run(contigs_paf, unitigs_path, nanopore_path, out_dir, threads=None, wiggle_room=300, device=0, timings=None,
        batches=0):
    """-> dict of counts; writes the three output files into out_dir (created by the caller, Application.cpp:65-82)."""
    t = {}
    n_threads = int(threads) if threads else max(1, min(16, os.cpu_count() or 1))
    params = overlap.default_params()
    params.wiggle_room = int(wiggle_room)

    # SequenceAccessor::buildIndex: parsing the two sequence files is pure host work that needs nothing from the PAF -- one
    # thread per file, started before the PAF is read, beside the parser, the GPU and the graph stage (ctypes calls release
    # the GIL).  Registry::operator[] for their records (SequenceAccessor.cpp:171,215) follows once the PAF's registries
    # exist.  Every HIP call stays on this thread: the upload happens after the join.
    seq = {}

    def parse_file(key, path):
        t1 = time.perf_counter()
        try:
            seq[key] = SeqFile(path)
        except BaseException as e:  # re-raised on the main thread
            seq["error"] = e
        t["sequences_parse_" + key] = time.perf_counter() - t1

    loaders = [threading.Thread(target=parse_file, args=("nanopore", nanopore_path), name="msgpu-nanopore"),
               threading.Thread(target=parse_file, args=("unitigs", unitigs_path), name="msgpu-unitigs")]
    for th in loaders:
        th.start()

    t0 = time.perf_counter()
    paf = overlap.parse_paf(contigs_paf, params)
    t["parse_paf"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    ctx = overlap.OverlapContext(device=device, params=params)
    ctx.set_id_space(paf.n_reads, paf.n_anchors)
    # the ThreadPool replacement (msgpu_overlap_batched_ex): rows -> HBM once, windows of owner reads on two streams, the
    # edge / order / id tables arrive in pinned host memory while later windows compute.  The EdgeMatch table (5/6 of the
    # bytes) stays in HBM: the graph stage never reads it, assemblePath only the path edges' (fetched below).
    tables, _ = ctx.overlap_batched(paf.rows, batches, copy=False, resident=True, edgematches=False)
    counts = ctx.counts()
    t["overlap_gpu"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    contraction = ctx.find_contraction_edges()
    t["contraction_gpu"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    g = GraphStage(tables, tables["read_len"], tables["read_first_line"])
    g.clean_up(contraction, paf.rows)
    g.linearize(n_threads)
    t["graph_host"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    g.set_path_edgematches(*ctx.get_edgematches(g.path_edges(), copy=False))  # MatchMap::getEdgeMatches of the path edges
    t["path_edgematches"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    for th in loaders:
        th.join()
    t["sequences_wait"] = time.perf_counter() - t0
    if "error" in seq:
        raise seq["error"]
    t0 = time.perf_counter()
    fn, fi = seq["nanopore"], seq["unitigs"]
    ids = (paf.register_sequences(NANOPORE, fn), paf.register_sequences(ILLUMINA, fi))
    t["sequences_registry"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    store = SeqStore(device=device)
    for kind, f, (ids_k, n) in zip((NANOPORE, ILLUMINA), (fn, fi), ids):
        store.upload(kind, f, ids_k, n)
    store.pack()  # 2 bits per base + exception list: a quarter of the footprint, less gather traffic, same bytes out
    t["sequences_upload"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    asm = Assembly(store)
    asm.set_rows(paf.rows, copy=False)  # (msgpu_assembly_borrow_rows: the loader's table outlives the layout)
    status = asm.add_prepared_batch([(g.path_input(i), g) for i in range(g.path_count)], n_threads) \
        if g.path_count else np.zeros(0, dtype=np.int32)
    st = g.stats
    g.close()
    ctx.close()  # (the graph had borrowed the context's pinned result tables: the context outlives it)
    asm.finish()
    t["assemble"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    for which, name in ((0, "temp_1.target.fa"), (1, "temp_1.query.fa"), (2, "temp_1.align.paf")):
        with open(os.path.join(out_dir, name), "wb") as f:
            f.write(asm.text(which))
    t["write"] = time.perf_counter() - t0
    info = asm.paths
    out = {"rows": len(paf.rows), "reads": int(counts.n_reads), "anchors": int(counts.n_anchors),
           "edges": int(counts.n_edges), "orders": int(counts.n_orders),
           "contraction_edges": int((contraction >= 0).sum()), "vertices_after_cleanup": int(st.n_vertices),
           "edges_after_cleanup": int(st.n_edges), "components": int(st.n_components), "paths": int(st.n_paths),
           "paths_skipped": int((status != 0).sum()), "contigs": int(len(info)),
           "target_bases": int(info["target_len"].sum()) if len(info) else 0, "queries": int(len(asm.queries))}
    if timings is not None:
        timings.update(t)
    store.close()
    return out
