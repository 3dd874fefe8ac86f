"""Full-size parity: the configurations BASELINE.json names (cfg2 = configs[1], cfg3 = configs[2], the one the
metric is quoted on; cfg4 = cfg3 cut into 8 shards) through the C-ABI on the GPU, compared BIT FOR BIT with the CPU
oracle on the same rows -- all four tables (edges, EdgeMatches, EdgeOrders, ids).  The C oracle needs ~3 s for cfg2 and
~40 s for cfg3 on one host core.  The consensus half at the same sizes: assemblePath over chains that tile the whole
genome, byte for byte against the Python restatement of ap.cpp."""
import os

import numpy as np
import pytest

from helpers import A9_TOLERANCE_VS_RESTATEMENT, assert_tables_equal

pytestmark = pytest.mark.gpu

_CACHE = {}


def _workload(cfg, oracle):
    """rows + oracle tables of a named configuration (kept for the module: cfg3's oracle run is the slow part)."""
    if cfg not in _CACHE:
        from muchsalsa_amd import synth
        tab = synth.paf_table(**synth.CONFIGS[cfg])
        rows, read_names, anchor_names = synth.accepted_rows(tab)
        _CACHE[cfg] = (rows, read_names, anchor_names, oracle.overlap(rows))
    return _CACHE[cfg]


def _shard_view(t, shard, world):
    """Vectorised statement of what msgpu_set_shard(shard, world) must return: the edges with v1 % world == shard cut
    out of the single-GPU tables, dense and re-based (muchsalsa_amd.distributed.shard_view_host without the loops)."""
    e, em, o, ids = t["edges"], t["ems"], t["orders"], t["ids"]
    keep = (e["v1"] % world) == shard
    e2 = e[keep].copy()
    em_keep = np.repeat(keep, e["em_cnt"])
    o_keep = np.repeat(keep, e["order_cnt"])
    em2, o2 = em[em_keep].copy(), o[o_keep].copy()
    id_keep = np.repeat(o_keep, o["ids_cnt"])
    ids2 = ids[id_keep].copy()
    new_edge = np.cumsum(keep) - 1
    em2["edge_idx"] = new_edge[em2["edge_idx"]]
    o2["edge_idx"] = new_edge[o2["edge_idx"]]
    e2["em_off"] = np.concatenate([[0], np.cumsum(e2["em_cnt"])[:-1]])
    e2["order_off"] = np.concatenate([[0], np.cumsum(e2["order_cnt"])[:-1]])
    o2["ids_off"] = np.concatenate([[0], np.cumsum(o2["ids_cnt"])[:-1]])
    return {"edges": e2, "ems": em2, "orders": o2, "ids": ids2}


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_named_config_bit_exact(oracle, cfg):
    """BASELINE.json configs[1] / configs[2] at full size: every field of every record equals the oracle's."""
    from muchsalsa_amd import overlap
    rows, read_names, anchor_names, want = _workload(cfg, oracle)
    with overlap.OverlapContext(0) as ctx:
        ctx.set_id_space(len(read_names), len(anchor_names))  # what bench.py does
        ctx.load_rows(rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        got = ctx.tables()
        c = ctx.counts()
        rl, fl = ctx.reads()
    assert_tables_equal(got, want, cfg)
    assert c.n_edges == len(want["edges"]) and c.n_ems == len(want["ems"]) and c.n_orders == len(want["orders"])
    assert np.array_equal(rl, want["read_len"]) and np.array_equal(fl, want["read_first_line"])
    if cfg == "cfg2":  # the shape SURVEY.md section 8 measured with the reference (different PRNG, same densities)
        assert 90_000 < c.n_edges < 105_000 and 2.4e6 < c.n_ems < 2.8e6
    else:
        assert 0.9e6 < c.n_edges < 1.05e6 and 24e6 < c.n_ems < 28e6


def test_cfg4_shards_of_cfg3_bit_exact(oracle):
    """BASELINE.json configs[3] (cfg3 sharded over 8 GPUs) as far as one GPU can show it: each of the 8 shards, run
    here one after the other, equals its cut of the oracle's cfg3 tables bit for bit; the cuts partition the edge list.
    (The all-gather + merge of the shards is covered at small size by test_gpu_parity / test_distributed_gloo.)"""
    from muchsalsa_amd import overlap
    rows, read_names, anchor_names, want = _workload("cfg3", oracle)
    world, n_edges = 8, 0
    with overlap.OverlapContext(0) as ctx:
        ctx.set_id_space(len(read_names), len(anchor_names))
        for r in range(world):
            ctx.set_shard(r, world)
            ctx.load_rows(rows)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            got = ctx.tables()
            assert_tables_equal(got, _shard_view(want, r, world), "cfg3 shard %d/8" % r)
            n_edges += len(got["edges"])
    assert n_edges == len(want["edges"])


def _sequences(cfg, read_names, anchor_names):
    """The synthetic genome's reads / unitigs keyed by Registry id, as the bytes the FASTA files hold."""
    from muchsalsa_amd import synth
    c = synth.CONFIGS[cfg]
    n_reads, L, seed = c["n_reads"], c["read_len"], c["seed"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, seed)
    a_start, a_len = synth.anchor_layout(n_reads, L, c["n_anchors"], seed)
    genome = synth.genome_bases(G, seed).tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    ro = np.array([int(n[1:]) for n in read_names])
    ao = np.array([int(n[1:]) for n in anchor_names])
    nano = {i: (genome[r_start[o]:r_start[o] + L] if r_fwd[o] else genome[r_start[o]:r_start[o] + L].translate(comp)[::-1])
            for i, o in enumerate(ro)}
    illu = {i: genome[a_start[o]:a_start[o] + a_len[o]] for i, o in enumerate(ao)}
    return G, L, r_start[ro], r_fwd[ro], nano, illu, genome


@pytest.mark.parametrize("cfg,two_bit", [("cfg2", False), ("cfg2", True), ("cfg3", True)])
def test_assemble_path_whole_genome(oracle, cfg, two_bit, tmp_path):
    """The consensus half at full size: chains of reads tiling the WHOLE synthetic genome (window = G) over the overlap
    tables the GPU just produced -> assemblePath (host layout on all host threads, one gather + FASTA wrapping on the
    device) -> temp_1.{target.fa, query.fa, align.paf} byte-identical to oracle/ms_assemble_py.py on the same input.
    A9 tolerance ("canonical order"): edit distance 0 against the restatement."""
    from oracle.ms_assemble_py import assemble_path
    from muchsalsa_amd import overlap, synth
    from muchsalsa_amd.assembly import Assembly
    from muchsalsa_amd.sequences import ILLUMINA, NANOPORE, SeqFile, SeqStore
    rows, read_names, anchor_names, want_tables = _workload(cfg, oracle)
    tables = overlap.build_overlaps(rows)
    assert tables["orders"].tobytes() == want_tables["orders"].tobytes()
    G, L, rs, rf, nano, illu, genome = _sequences(cfg, read_names, anchor_names)
    paths = synth.chain_paths(tables, rs, rf, L, G, max_reads=12)
    assert sum(len(p) for p, _ in paths) > 0.1 * len(read_names)
    for name, seqs in (("n.fa", nano), ("i.fa", illu)):
        with open(tmp_path / name, "wb") as f:
            for i in range(len(seqs)):
                f.write(b">s%d\n" % i + seqs[i] + b"\n")
    store = SeqStore(device=0)
    store.upload(NANOPORE, SeqFile(str(tmp_path / "n.fa")))
    store.upload(ILLUMINA, SeqFile(str(tmp_path / "i.fa")))
    os.remove(tmp_path / "n.fa")
    if two_bit:
        store.pack()
    asm = Assembly(store)
    asm.set_rows(rows)
    prepared = [Assembly.prepare(p, st, None, None, k) for k, (p, st) in enumerate(paths)]
    status = asm.add_prepared_batch(prepared, min(16, os.cpu_count() or 1))
    assert not status.any()
    asm.finish()
    # the restatement needs (read, anchor) -> row for the reads on paths only
    on_path = np.zeros(len(read_names), dtype=bool)
    for p, _ in paths:
        on_path[[r["id"] for r in p]] = True
    vm = {(int(r["read_id"]), int(r["anchor_id"])): r for r in rows[on_path[rows["read_id"]]]}
    want = [assemble_path(p, st, vm, {}, nano, illu, k) for k, (p, st) in enumerate(paths)]
    assert A9_TOLERANCE_VS_RESTATEMENT == 0  # the stated tolerance: identical texts
    assert asm.text(2) == b"".join(r["paf"] for r in want)
    assert asm.text(0) == b"".join(r["target_fa"] for r in want)
    assert asm.text(1) == b"".join(r["query_fa"] for r in want)
    T = int(asm.paths["target_len"].sum())
    assert T == sum(len(r["target"]) for r in want) and T > 0.9 * G  # the chains tile the genome
    store.close()


# ---- the rows SURVEY.md section 8(f) marks "next", at full size: findContractionEdges (GPU), graph stage, real-path A9 ----

def _tiled(cfg, oracle):
    """synth.TILED[cfg]: unitigs that tile the genome, reads of mixed length -> rows, names, oracle tables"""
    key = "tiled-" + cfg
    if key not in _CACHE:
        from muchsalsa_amd import synth
        rows, read_names, anchor_names = synth.accepted_rows(synth.paf_table(**synth.TILED[cfg]))
        _CACHE[key] = (rows, read_names, anchor_names, oracle.overlap(rows))
    return _CACHE[key]


def _promote_contained(t):
    """every contained EdgeOrder made primary: more candidates for findContractionEdges, all four containment cases of
    sanityCheck (sc.cpp:41-82) at scale"""
    o = t["orders"].copy()
    o["flags"] |= np.where(o["flags"] & 2, 8, 0).astype(np.uint32)
    return dict(t, orders=o)


@pytest.mark.parametrize("shape,cfg", [("baseline", "cfg2"), ("baseline", "cfg3"), ("tiled", "cfg2"), ("tiled", "cfg3")])
def test_find_contraction_edges_full_size(oracle, shape, cfg):
    """findContractionEdges + sanityCheck (src/main.cpp:416-463, sc.cpp:29-90) as k_check_contraction at the size of
    BASELINE.json configs[1] / configs[2]: on the tables the GPU just produced (bit-exact themselves), plain and with
    every contained order promoted to primary, == the C oracle's list.  The BASELINE shape has no contraction edge until
    promoted (reads of one length are never contained); the tiled shape has tens of thousands."""
    import torch
    from muchsalsa_amd import overlap
    rows, read_names, anchor_names, want = (_workload if shape == "baseline" else _tiled)(cfg, oracle)
    n = len(want["read_len"])
    with overlap.OverlapContext(0) as ctx:
        ctx.set_id_space(len(read_names), len(anchor_names))
        ctx.load_rows(rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        assert_tables_equal(ctx.tables(), want, "%s %s" % (shape, cfg))
        got = ctx.find_contraction_edges()
        want_co = oracle.find_contraction_edges(want, n)
        assert np.array_equal(got, want_co)
        hits = int((want_co >= 0).sum())
        assert shape == "baseline" or hits > len(want["edges"]) // 10, hits
        promoted = _promote_contained(want)
        d_e = torch.from_numpy(promoted["edges"].view(np.uint8).copy()).cuda()
        d_o = torch.from_numpy(promoted["orders"].view(np.uint8).copy()).cuda()
        got = ctx.find_contraction_edges(d_e.data_ptr(), len(promoted["edges"]), d_o.data_ptr(), len(promoted["orders"]), n)
        want_co = oracle.find_contraction_edges(promoted, n)
        assert np.array_equal(got, want_co)
        n_contained = int(((want["orders"]["flags"] & 2) != 0).sum())
        if n_contained:
            assert int((want_co >= 0).sum()) >= hits and int((want_co >= 0).sum()) > 0
        print("%s %s: %d edges, %d contained orders, %d contraction edges, %d when promoted" % (
            shape, cfg, len(want["edges"]), n_contained, hits, int((want_co >= 0).sum())))


@pytest.mark.parametrize("shape,cfg", [("baseline", "cfg2"), ("tiled", "cfg3")])
def test_graph_stage_full_size_on_gpu_tables(oracle, shape, cfg):
    """The flat-CSR host graph stage (rooted span forest for decycle, per-component heap for extractPaths) on the tables
    and the contraction list the GPU produced, == oracle/ms_graph_py.py: BASELINE.json configs[1] (98 k edges, one giant
    component of shadow edges) and the tiled shape at the size of configs[2] (100 k reads, 578 k edges, 94 k contraction
    edges, hundreds of components); configs[1] also with one order direction in 40 flipped (12 k decycle conflicts)."""
    from test_graph_fullsize import compare_stage, flip_strands
    from muchsalsa_amd import overlap
    rows, read_names, anchor_names, want = (_workload if shape == "baseline" else _tiled)(cfg, oracle)
    n = len(want["read_len"])
    with overlap.OverlapContext(0) as ctx:
        ctx.set_id_space(len(read_names), len(anchor_names))
        t, _ = ctx.overlap_batched(rows, 4, resident=True)
        assert_tables_equal(t, want, "%s %s" % (shape, cfg))
        co = ctx.find_contraction_edges()
        c = compare_stage(oracle, rows, t, co, threads=8)
        print(shape, cfg, c)
        assert c is not None and c["n_paths"] > 0
        if shape == "tiled":
            assert c["n_contraction_edges"] > 50_000 and c["n_components"] > 100 and c["longest_path"] > 200
        if shape == "tiled":  # (the flipped variant of the tiled shape runs at cfg2 size in tests/test_graph_fullsize.py)
            return
        t2 = flip_strands(t, 11, every=40)
        d_e = __import__("torch").from_numpy(t2["edges"].view(np.uint8).copy()).cuda()
        d_o = __import__("torch").from_numpy(t2["orders"].view(np.uint8).copy()).cuda()
        co2 = ctx.find_contraction_edges(d_e.data_ptr(), len(t2["edges"]), d_o.data_ptr(), len(t2["orders"]), n)
        assert np.array_equal(co2, oracle.find_contraction_edges(t2, n))
    c2 = compare_stage(oracle, rows, t2, co2, threads=8)
    print(shape, cfg, "flipped:", c2)
    assert c2 is None or c2["n_decycled_edges"] > 0


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_whole_flow_tiled_full_size(oracle, cfg, tmp_path):
    """PAF text + FASTA in -> temp_1.{target.fa, query.fa, align.paf} out of muchsalsa_amd.pipeline.run (the dispatcher
    with the EdgeMatch table left in HBM, the GPU contraction test, the host graph stage, the path edges' EdgeMatches on
    demand, assemblePath over the paths linearizeGraph yields) at the size of configs[1] / configs[2] on the tiled shape,
    byte-identical to the flow made of oracles only (C overlap oracle, C findContractionEdges, Python graph stage, Python
    assemblePath): A9 on the REAL paths, tolerance 0 against the restatement -- and the contigs cover the genome."""
    from graphcases import synth_sequences
    from test_gpu_pipeline import oracle_flow
    from muchsalsa_amd import pipeline, synth
    shape = synth.TILED[cfg]
    tab = synth.paf_table(**shape)
    rows, read_names, anchor_names, _ = _tiled(cfg, oracle)
    G, rs, rf, rl, nano, illu = synth_sequences(shape, read_names, anchor_names)
    (tmp_path / "contigs.paf").write_text("\n".join(synth.paf_lines(tab)) + "\n")
    with open(tmp_path / "unitigs.fa", "wb") as f:
        for j, name in enumerate(anchor_names):
            f.write(b">%s\n" % name.encode() + illu[j] + b"\n")
    with open(tmp_path / "nanopore.fa", "wb") as f:
        for i, name in enumerate(read_names):
            f.write(b">%s\n" % name.encode() + nano[i] + b"\n")
    out = tmp_path / "out"
    out.mkdir()
    timings = {}
    res = pipeline.run(str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / "nanopore.fa"), str(out),
                       threads=min(16, os.cpu_count() or 1), timings=timings)
    print(cfg, res, {k: round(v, 4) for k, v in timings.items()})
    assert res["rows"] == len(rows) and res["paths_skipped"] == 0 and res["contraction_edges"] > res["edges"] // 10
    want = oracle_flow(oracle, rows, nano, illu)
    assert res["contigs"] == len(want) > 0
    assert A9_TOLERANCE_VS_RESTATEMENT == 0
    assert (out / "temp_1.align.paf").read_bytes() == b"".join(r["paf"] for r in want)
    assert (out / "temp_1.target.fa").read_bytes() == b"".join(r["target_fa"] for r in want)
    assert (out / "temp_1.query.fa").read_bytes() == b"".join(r["query_fa"] for r in want)
    assert 0.97 * G < res["target_bases"] < 1.03 * G  # (contigs tile the genome; joints duplicate a few bases each)
