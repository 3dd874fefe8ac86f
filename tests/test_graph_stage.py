"""Host graph stage (clean-up, span tree, decycle, components, getDirectedGraph, linearizeGraph; src/main.cpp:194-310,
mst.cpp, cc.cpp, dg.cpp, lg.cpp) of libmsgpu against the object-model restatement in oracle/ms_graph_py.py.  No GPU:
the contraction edges come from the C oracle here (tests/test_gpu_pipeline.py takes them from the HIP kernel)."""
import numpy as np
import pytest

from graphcases import varlen_rows

from muchsalsa_amd import _lib
from muchsalsa_amd.graph import GraphStage
from muchsalsa_amd.overlap import MsgpuError


def oracle_stage(oracle, rows, t, co):
    from oracle import ms_graph_py as G
    vm = {(int(r["read_id"]), int(r["anchor_id"])) for r in rows}
    g, el, oo = G.build_graph(t, t["read_len"], t["read_first_line"])
    contain = G.clean_up(g, el, oo, co, lambda r, a: (r, a) in vm)
    state = dict(vertex_alive=np.array([v in g.vertices for v in range(len(t["read_len"]))]),
                 edge_alive=np.array([id(e) in g.edges for e in el]),
                 edge_consensus=np.array([{G.POS: 1, G.NEG: 0, G.NONE: 2}[e.consensus] for e in el], dtype=np.uint8),
                 edge_weight=np.array([e.weight for e in el], dtype=np.uint64))
    edges, ems = t["edges"], t["ems"]
    eidx = {(int(e["v1"]), int(e["v2"])): i for i, e in enumerate(edges)}

    def em_of(a, b):
        e = edges[eidx[(a, b)]]
        return {int(m["anchor_id"]): (int(m["ov_lo"]), int(m["ov_hi"]))
                for m in ems[int(e["em_off"]): int(e["em_off"]) + int(e["em_cnt"])]}
    paths = G.assemble_all(g, em_of)
    dirs = np.array([{G.POS: 1, G.NEG: 0, G.NONE: 2}[g.vertices[v].direction] if v in g.vertices else 255
                     for v in range(len(t["read_len"]))])
    return state, contain, paths, dirs


def variants(oracle):
    """(name, rows, tables, contraction_order): random and tiled anchors, natural and forced contractions, mixed shadows"""
    for seed, tiled in ((1, False), (2, True), (3, True), (4, False)):
        rows = varlen_rows(400, 2500, 250_000, seed, tiled=tiled)
        t = oracle.overlap(rows)
        n = len(t["read_len"])
        yield "seed%d%s" % (seed, "t" if tiled else "r"), rows, t, oracle.find_contraction_edges(t, n)
        t2 = dict(t, edges=t["edges"].copy(), orders=t["orders"].copy())
        t2["orders"]["flags"] |= np.where(t2["orders"]["flags"] & 2, 8, 0).astype(np.uint32)  # contained -> primary
        if seed & 1:
            t2["edges"]["shadow"] ^= (np.arange(len(t2["edges"])) % 3 == 0).astype(np.uint8)
        yield "seed%d-forced" % seed, rows, t2, oracle.find_contraction_edges(t2, n)
        # inconsistent strands: flip EdgeOrder::direction on a tenth of the orders -> odd cycles for decycle(), shadow
        # edges whose orders disagree (consensus e_NONE), vertices getDirectedGraph never orients
        t3 = dict(t, edges=t["edges"].copy(), orders=t["orders"].copy())
        rng = np.random.default_rng(seed)
        t3["orders"]["flags"] ^= np.where(rng.integers(0, 10, len(t3["orders"])) == 0, 4, 0).astype(np.uint32)
        t3["edges"]["shadow"] ^= (rng.integers(0, 4, len(t3["edges"])) == 0).astype(np.uint8)
        yield "seed%d-strands" % seed, rows, t3, oracle.find_contraction_edges(t3, n)


def test_graph_stage_matches_oracle(oracle):
    seen = dict(contain=0, decycled=0, deleted=0, paths=0, long_path=0, multi_comp=0)
    for name, rows, t, co in variants(oracle):
        from oracle.ms_graph_py import GraphError
        try:
            state, contain, paths, dirs = oracle_stage(oracle, rows, t, co)
        except GraphError:
            state = None
        g = GraphStage(t, t["read_len"], t["read_first_line"])
        if state is None:
            with pytest.raises(MsgpuError):
                g.clean_up(co, rows)
                g.linearize()
            continue
        g.clean_up(co, rows)
        got = g.state()
        for k in ("vertex_alive", "edge_alive"):
            assert np.array_equal(got[k], state[k]), (name, k)
        alive = state["edge_alive"]
        assert np.array_equal(got["edge_consensus"][alive], state["edge_consensus"][alive]), name
        assert np.array_equal(got["edge_weight"][alive], state["edge_weight"][alive]), name
        g.linearize(threads=1 + len(name) % 5)  # the thread count must not matter
        st = g.stats
        assert st.n_paths == len(paths), name
        got_contain = {}
        for i, (want_path, want_steps) in enumerate(paths):
            p, steps, cont = g.path(i)
            assert p == want_path, (name, i)
            assert steps == want_steps, (name, i)
            got_contain.update(cont)
        on_paths = {r["id"] for p, _ in paths for r in p}
        want_contain = {v: [dict(nano=c["nano"], dir=c["direction"], anchors=c["anchors"]) for c in lst]
                        for v, lst in contain.items() if v in on_paths}
        assert got_contain == want_contain, name
        vd = g.state()["vertex_direction"]
        assert np.array_equal(vd[state["vertex_alive"]], dirs[state["vertex_alive"]]), name
        assert st.n_contain_elements == sum(len(v) for v in contain.values())
        seen["contain"] += st.n_contain_elements
        seen["decycled"] += st.n_decycled_edges
        seen["deleted"] += st.n_deleted_vertices
        seen["paths"] += st.n_paths
        seen["long_path"] += any(len(p) >= 10 for p, _ in paths)
        # every vertex of a component is reached through edges with a consensus direction, so none stays e_NONE
        assert not (dirs[state["vertex_alive"]] == 2).any()
        seen["multi_comp"] += st.n_components > 1
    for k, v in seen.items():
        assert v > 0, "never exercised: %s (%r)" % (k, seen)


def test_graph_stage_argument_and_state_errors(oracle):
    rows = varlen_rows(60, 300, 40_000, 5, tiled=True)
    t = oracle.overlap(rows)
    g = GraphStage(t, t["read_len"], t["read_first_line"])
    with pytest.raises(MsgpuError) as e:
        g.linearize()  # before clean_up
    assert e.value.code == _lib.E_STATE
    bad = np.full(len(t["edges"]), -1, dtype=np.int64)
    if len(t["orders"]):
        bad[0] = len(t["orders"]) + 5  # not an order of that edge
        with pytest.raises(MsgpuError) as e:
            g.clean_up(bad)
        assert e.value.code == _lib.E_ARG
    g.clean_up(np.full(len(t["edges"]), -1, dtype=np.int64))
    with pytest.raises(MsgpuError) as e:
        g.clean_up(np.full(len(t["edges"]), -1, dtype=np.int64))
    assert e.value.code == _lib.E_STATE
    g.linearize()
    assert g.path_count == g.stats.n_paths
    t_bad = dict(t, edges=t["edges"].copy())
    t_bad["edges"]["v2"][0] = 10 ** 6  # vertex out of range
    with pytest.raises(MsgpuError):
        GraphStage(t_bad, t["read_len"], t["read_first_line"])


def random_tables(rng, n_reads, n_edges, max_orders=3):
    """Arbitrary (not overlap-derived) edge/order tables: random pairs, 1..max_orders orders per edge with random start
    side, strand, containment, scores; a few anchors per order.  Everything the graph stage reads, nothing it checks."""
    from muchsalsa_amd._lib import EDGE_DTYPE, EM_DTYPE, ORDER_DTYPE
    pairs = set()
    while len(pairs) < n_edges:
        a, b = int(rng.integers(0, n_reads)), int(rng.integers(0, n_reads))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    pairs = sorted(pairs)
    edges = np.zeros(len(pairs), dtype=EDGE_DTYPE)
    orders, ids, ems = [], [], []
    for i, (a, b) in enumerate(pairs):
        k = int(rng.integers(1, max_orders + 1))
        edges[i] = (a, b, len(ems), len(orders), 2, k, int(rng.integers(0, 4) == 0), 0)
        anchors = [int(x) for x in rng.choice(500, 2, replace=False)]
        for anc in anchors:
            ems.append((10, 400, 1.0, anc, 0, 3, i))
        for _ in range(k):
            start_v1 = bool(rng.integers(0, 2))
            fl = (1 if start_v1 else 0) | (2 if rng.integers(0, 5) == 0 else 0) | (4 if rng.integers(0, 3) else 0) | \
                 (8 if rng.integers(0, 2) else 0)
            orders.append((i, fl, float(rng.integers(0, 400)), float(rng.integers(0, 400)), int(rng.integers(500, 5000)),
                           len(ids), 2, a if start_v1 else b, b if start_v1 else a, a, (0, 0)))
            ids.extend(anchors)
    return dict(edges=edges, ems=np.array(ems, dtype=EM_DTYPE), orders=np.array(orders, dtype=ORDER_DTYPE),
                ids=np.array(ids, dtype="<u4"), read_len=rng.integers(1000, 9000, n_reads).astype("<i4"),
                read_first_line=rng.permutation(n_reads).astype("<u4"))


def test_graph_stage_on_random_graphs(oracle):
    """Differential fuzz on graphs the overlap path would never produce: dense cycles, two-way directed edges, strand
    conflicts, many contractions.  Product and object-model oracle must agree on every state, path and rejection."""
    from oracle import ms_graph_py as G
    rng = np.random.default_rng(2024)
    G.COVER.clear()
    n_ok = n_rejected = n_paths = 0
    for trial in range(400):
        n_reads = int(rng.integers(5, 90))
        n_edges = int(rng.integers(n_reads // 2, min(n_reads * 3, n_reads * (n_reads - 1) // 2) + 1))
        t = random_tables(rng, n_reads, n_edges)
        co = oracle.find_contraction_edges(t, n_reads)
        rows = np.zeros(0, dtype=_lib.ROW_DTYPE)
        try:
            g, el, oo = G.build_graph(t, t["read_len"], t["read_first_line"])
            contain = G.clean_up(g, el, oo, co, lambda r, a: True)
            state_alive = np.array([id(e) in g.edges for e in el])
            cons = np.array([{G.POS: 1, G.NEG: 0, G.NONE: 2}[e.consensus] for e in el], dtype=np.uint8)
            paths = G.assemble_all(g, lambda a, b: {})
        except (G.GraphError, KeyError, IndexError):
            paths = None
        gs = GraphStage(t, t["read_len"], t["read_first_line"])
        if paths is None:
            with pytest.raises(MsgpuError):
                gs.clean_up(co, None)
                gs.linearize()
            n_rejected += 1
            continue
        gs.clean_up(co, None)
        st = gs.state()
        assert np.array_equal(st["edge_alive"], state_alive), trial
        assert np.array_equal(st["edge_consensus"][state_alive], cons[state_alive]), trial
        gs.linearize()
        assert gs.path_count == len(paths), trial
        for i, (want_path, want_steps) in enumerate(paths):
            p, steps, cont = gs.path(i)
            assert p == want_path, (trial, i)
            assert [s["orders"] for s in steps] == [s["orders"] for s in want_steps], (trial, i)
        n_ok += 1
        n_paths += len(paths)
    print('random graphs: ok %d, rejected %d, paths %d, branches %r' % (n_ok, n_rejected, n_paths, G.COVER))
    for k in ("cycle_cut", "decycle", "decycle_weak_tree_edge", "short_path_dropped", "two_way_edge"):
        assert G.COVER.get(k, 0) > 0, "branch never reached: " + k
    assert n_ok > 200 and n_paths > 300, (n_ok, n_rejected, n_paths)


def _host_edgematches_of(t, idx):
    """what msgpu_get_edgematches returns for a list of edge-table indices, stated on the host tables"""
    e = t["edges"][idx]
    off = np.concatenate([[0], np.cumsum(e["em_cnt"].astype(np.uint64))]).astype("<u8")
    parts = [t["ems"][int(o): int(o) + int(c)] for o, c in zip(e["em_off"], e["em_cnt"])]
    return off, (np.concatenate(parts) if parts else np.zeros(0, dtype=_lib.EM_DTYPE))


def test_edgematches_on_demand_give_the_same_path_inputs(oracle):
    """A graph created WITHOUT the EdgeMatch table (it stays in HBM in the product flow) + the path edges' EdgeMatches
    supplied after linearize == the graph created with the whole table (MatchMap::getEdgeMatches is only ever asked for
    path edges: dg.cpp:99-101 -> ap.cpp:631-706)."""
    rows = varlen_rows(400, 2500, 250_000, 2, tiled=True)
    t = oracle.overlap(rows)
    co = oracle.find_contraction_edges(t, len(t["read_len"]))
    full = GraphStage(t, t["read_len"], t["read_first_line"])
    lean = GraphStage(dict(t, ems=None), t["read_len"], t["read_first_line"])
    for g in (full, lean):
        with pytest.raises(MsgpuError) as e:
            g.path_edges()
        assert e.value.code == _lib.E_STATE
        g.clean_up(co, rows)
        g.linearize(3)
    assert full.path_count == lean.path_count > 0
    with pytest.raises(MsgpuError) as e:
        lean.path_input(0)
    assert e.value.code == _lib.E_STATE
    idx = lean.path_edges()
    assert np.array_equal(idx, full.path_edges())
    assert len(idx) == sum(full.path_input(i).n_reads - 1 for i in range(full.path_count))
    off, ems = _host_edgematches_of(t, idx)
    with pytest.raises(MsgpuError) as e:  # a list that does not match the edges' EdgeMatch counts is refused
        bad = off.copy()
        bad[1:] += 1
        lean.set_path_edgematches(bad, np.concatenate([ems, ems[:1]]))
    assert e.value.code == _lib.E_ARG
    lean.set_path_edgematches(off, ems)
    for i in range(full.path_count):
        assert lean.path(i) == full.path(i), i


def test_graph_create_copies_and_create_borrowed_borrows(oracle):
    """msgpu_graph_create owns copies of the tables (the caller may free or reuse its buffers right away);
    msgpu_graph_create_borrowed is the variant that reads the caller's memory until msgpu_graph_free."""
    import ctypes as C
    L = _lib.lib()
    rows = varlen_rows(200, 1500, 120_000, 6, tiled=True)
    t = oracle.overlap(rows)
    co = oracle.find_contraction_edges(t, len(t["read_len"]))
    ref = GraphStage(t, t["read_len"], t["read_first_line"])
    ref.clean_up(co, rows)
    ref.linearize()
    want = [ref.path(i) for i in range(ref.path_count)]
    assert want
    e, m, o, i = (t[k].copy() for k in ("edges", "ems", "orders", "ids"))
    rl, fl = t["read_len"].astype("<i4"), t["read_first_line"].astype("<u4")
    h = C.c_void_p()
    assert L.msgpu_graph_create(e.ctypes.data, len(e), m.ctypes.data, len(m), o.ctypes.data, len(o), i.ctypes.data, len(i),
                                rl.ctypes.data, fl.ctypes.data, len(rl), C.byref(h)) == 0
    for a in (e, m, o, i):  # scramble the caller's buffers: the graph must not notice
        a.view(np.uint8)[:] = 0xEE
    cov = np.ascontiguousarray(co, dtype="<i8")
    assert L.msgpu_graph_clean_up(h, cov.ctypes.data, None, 0) == 0
    assert L.msgpu_graph_linearize(h) == 0
    assert L.msgpu_graph_path_count(h) == len(want)
    for k, (wp, _, _) in enumerate(want):
        p = _lib.PathInput()
        assert L.msgpu_graph_path_input(h, k, C.byref(p)) == 0
        reads = np.frombuffer(C.string_at(p.reads, p.n_reads * 16), dtype=_lib.PATH_READ_DTYPE)
        assert [int(r["read_id"]) for r in reads] == [r["id"] for r in wp]
        n_ids = max((int(x["ids_off"]) + int(x["ids_cnt"]) for x in np.frombuffer(
            C.string_at(p.orders, int(np.frombuffer(C.string_at(p.order_off, 4 * p.n_reads), "<u4")[-1]) * 24),
            dtype=_lib.PATH_ORDER_DTYPE)), default=0)
        if n_ids:  # the id pool the path input points into is the graph's own copy, not the scrambled buffer
            assert np.array_equal(np.frombuffer(C.string_at(p.ids, 4 * n_ids), "<u4"), t["ids"][:n_ids])
    L.msgpu_graph_free(h)
