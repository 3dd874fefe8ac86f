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
    # the GIL).  A file's bytes travel to HBM while it is parsed (msgpu_seq_parse_upload: the host never holds them) and are
    # converted to the 2-bit form there (the store has one stream per kind).  Registry::operator[] for the records (SequenceAccessor.cpp:171,215) follows on a
    # third thread once the PAF's registries exist, beside the overlap and graph stages.
    seq = {}
    store_ready = threading.Event()

    def parse_file(key, kind, path):
        try:
            store_ready.wait()
            if "store" not in seq:
                return
            t1 = time.perf_counter()
            seq[key] = seq["store"].parse_upload(kind, path)  # records -> page-locked ring -> HBM while the file is read
            t["sequences_parse_" + key] = time.perf_counter() - t1
            t1 = time.perf_counter()
            seq["store"].pack_store(kind)  # 2 bits per base + exception list: a quarter of the footprint, same bytes out
            t["sequences_pack_" + key] = time.perf_counter() - t1
        except BaseException as e:  # re-raised on the main thread
            seq["error"] = e

    loaders = [threading.Thread(target=parse_file, args=("nanopore", NANOPORE, nanopore_path), name="msgpu-nanopore"),
               threading.Thread(target=parse_file, args=("unitigs", ILLUMINA, unitigs_path), name="msgpu-unitigs")]
    def make_store():  # (the HIP runtime starts here, beside the PAF parser)
        t1 = time.perf_counter()
        try:
            seq["store"] = SeqStore(device=device)
        except BaseException as e:
            seq["error"] = e
        finally:
            store_ready.set()
        t["device_init"] = time.perf_counter() - t1

    for th in [threading.Thread(target=make_store, name="msgpu-device")] + loaders:
        th.start()

    t0 = time.perf_counter()
    paf = overlap.parse_paf(contigs_paf, params)
    t["parse_paf"] = time.perf_counter() - t0

    n_reads, n_anchors = paf.n_reads, paf.n_anchors  # (the sequence files' records may add ids to the registries)

    def register():
        try:
            for th in loaders:
                th.join()
            if "error" in seq:
                return
            t1 = time.perf_counter()
            store = seq["store"]

            def one(kind, key):  # (two registries, two stores: the kinds do not meet)
                try:
                    store.set_ids(kind, seq[key], *paf.register_sequences(kind, seq[key]))
                except BaseException as e:
                    seq["error"] = e

            other = threading.Thread(target=one, args=(ILLUMINA, "unitigs"), name="msgpu-registry-unitigs")
            other.start()
            one(NANOPORE, "nanopore")
            other.join()
            t["sequences_registry"] = time.perf_counter() - t1
        except BaseException as e:
            seq["error"] = e

    registrar = threading.Thread(target=register, name="msgpu-registry")
    registrar.start()

    t0 = time.perf_counter()
    ctx = overlap.OverlapContext(device=device, params=params)
    ctx.set_id_space(n_reads, n_anchors)
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
    registrar.join()
    t["sequences_wait"] = time.perf_counter() - t0
    if "error" in seq:
        raise seq["error"]
    store = seq["store"]

    t0 = time.perf_counter()
    asm = Assembly(store)
    asm.set_rows(paf.rows, copy=False)  # (msgpu_assembly_borrow_rows: the loader's table outlives the layout)
    status = asm.add_graph_paths(g, n_threads)  # the assemblePaths fan-out over the paths linearizeGraph yielded
    st = g.stats
    n_rows = len(paf.rows)
    held, held_rows = [g, ctx], [paf, tables]
    del paf, tables
    asm._rows_keep = None  # (the layout is done: nothing reads the borrowed row table any more)

    def release():
        # the graph and the overlap context (device tables, page-locked result tables: the graph had borrowed them, so it
        # goes first) are done with: they go back beside the gather and the writes ...
        held[0].close()
        held[1].close()
        del held[:]

    def release_rows():  # ... and so does the loader's page-locked row table (its last views die here)
        del held_rows[:]

    releasers = [threading.Thread(target=release, name="msgpu-release"),
                 threading.Thread(target=release_rows, name="msgpu-release-rows")]
    for th in releasers:  # (started behind the gather instead, the same work shows up as teardown: measured, no difference)
        th.start()
    asm.finish()
    t["assemble"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    failed = []

    def write_file(which, name):  # (write() releases the GIL: the three files go out side by side, straight from the library's buffers)
        try:
            with open(os.path.join(out_dir, name), "wb") as f:
                f.write(asm.text_view(which))
        except BaseException as e:
            failed.append(e)

    writers = [threading.Thread(target=write_file, args=a) for a in ((0, "temp_1.target.fa"), (1, "temp_1.query.fa"))]
    for th in writers:
        th.start()
    write_file(2, "temp_1.align.paf")
    for th in writers:
        th.join()
    if failed:
        raise failed[0]
    t["write"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    info = asm.paths
    out = {"rows": n_rows, "reads": int(counts.n_reads), "anchors": int(counts.n_anchors),
           "edges": int(counts.n_edges), "orders": int(counts.n_orders),
           "contraction_edges": int((contraction >= 0).sum()), "vertices_after_cleanup": int(st.n_vertices),
           "edges_after_cleanup": int(st.n_edges), "components": int(st.n_components), "paths": int(st.n_paths),
           "paths_skipped": int((status != 0).sum()), "contigs": int(len(info)),
           "target_bases": int(info["target_len"].sum()) if len(info) else 0, "queries": asm.query_count}
    t["collect"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    asm.close()
    store.close()
    for key in ("nanopore", "unitigs"):
        seq[key].close()
    for th in releasers:
        th.join()
    t["teardown"] = time.perf_counter() - t0
    if timings is not None:
        timings.update(t)
    return out
