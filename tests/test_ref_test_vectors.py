"""The vectors the REFERENCE'S OWN unit tests hold (libms/tests/*.cpp), replayed through the C-ABI and the oracles.

tests/golden/ref_tests/*.json are data (numbers and strings) extracted by tools/make_ref_test_fixtures.py; each names the
test file and lines it came from.  They are the part of the oracle's pin that is reference-held: what libms' maintainers
assert about getMaxSpanTree, getConnectedComponents, getShortestPath, sortTopologically, readline, Registry and Toggle
is asserted here about libmsgpu (same code paths the graph stage and the loader run) and about oracle/ms_graph_py.py /
oracle/ms_oracle.c / oracle/ms_oracle_py.py.  No GPU needed: these are host entry points."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests")


def fixture(name):
    with open(os.path.join(HERE, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    g.build()
    from muchsalsa_amd import _lib
    return _lib.lib()


def u32(x):
    return np.ascontiguousarray(x, dtype="<u4")


def _ograph(vertices, edges, directed=False):
    from oracle import ms_graph_py as G
    g = G.DiGraph() if directed else G.Graph()
    for v in vertices:
        g.add_vertex(G.Vertex(v, 0, 0))
    for a, b in edges:
        g.add_edge(a, b)
    return g, G


def test_max_span_tree_MST_test(L, oracle):
    fx = fixture("mst")
    assert fx["source"].startswith("libms/tests/MST_test.cpp")
    a, b = u32([e[0] for e in fx["edges"]]), u32([e[1] for e in fx["edges"]])
    w = np.ascontiguousarray([e[2] for e in fx["edges"]], dtype="<u8")
    n = max(fx["vertices"]) + 1  # ids are used as they stand; vertex 0 stays isolated
    in_tree = np.zeros(len(a), dtype=np.uint8)
    # before any consensus direction is set no edge qualifies (mst.cpp:79-86): size 0
    none = np.full(len(a), 2, dtype=np.uint8)
    assert L.msgpu_graph_max_span_tree(n, a.ctypes.data, b.ctypes.data, w.ctypes.data, none.ctypes.data, len(a),
                                       in_tree.ctypes.data) == 0
    assert int(in_tree.sum()) == fx["expect"]["size_before_consensus"]
    cons = np.full(len(a), 2, dtype=np.uint8)
    for x, y in fx["consensus_true"]:
        cons[[i for i, e in enumerate(fx["edges"]) if e[:2] == [x, y]][0]] = 1
    assert L.msgpu_graph_max_span_tree(n, a.ctypes.data, b.ctypes.data, w.ctypes.data, cons.ctypes.data, len(a),
                                       in_tree.ctypes.data) == 0
    tree = {frozenset(e[:2]) for i, e in enumerate(fx["edges"]) if in_tree[i]}
    assert len(tree) == fx["expect"]["size"]
    assert all(frozenset(p) in tree for p in fx["expect"]["has_edge"])
    assert not any(frozenset(p) in tree for p in fx["expect"]["has_no_edge"])
    # the restatement (object model)
    g, G = _ograph(fx["vertices"], [e[:2] for e in fx["edges"]])
    for x, y, wt in fx["edges"]:
        g.get_edge(x, y).weight = wt
    assert G.max_span_tree(g).size() == fx["expect"]["size_before_consensus"]
    for x, y in fx["consensus_true"]:
        g.get_edge(x, y).set_consensus(True)
    mst = G.max_span_tree(g)
    assert mst.size() == fx["expect"]["size"] and mst.order() == g.order()
    assert all(mst.has_edge(*p) for p in fx["expect"]["has_edge"])
    assert not any(mst.has_edge(*p) for p in fx["expect"]["has_no_edge"])


def test_max_span_tree_on_all_threads_is_kruskals_tree(L, monkeypatch):
    """getMaxSpanTree on a large graph finds the forest with Boruvka's rounds on all host threads (csrc/graph_stage.cpp):
    the same edges as the one-thread loop of mst.cpp:75-111 -- heaviest first, equal weights in edge order -- on random
    multigraphs with many equal and zero weights, loops, edges without a consensus direction and isolated vertices."""
    rng = np.random.default_rng(11)
    for case in range(40):
        n = int(rng.integers(2, 400))
        m = int(rng.integers(1, 6 * n))
        a, b = u32(rng.integers(0, n, m)), u32(rng.integers(0, n, m))
        if case % 3 == 0:  # long chains: many rounds, long hooks
            k = min(m, n - 1)
            a[:k], b[:k] = np.arange(k), np.arange(1, k + 1)
        w = np.ascontiguousarray(rng.integers(0, 4, m) * (rng.integers(0, 3, m) > 0), dtype="<u8")
        cons = np.ascontiguousarray(rng.integers(0, 3, m), dtype=np.uint8)  # 2 = e_NONE: not a candidate
        want, got = np.zeros(m, np.uint8), np.zeros(m, np.uint8)
        monkeypatch.delenv("MSGPU_GRAPH_PAR_MIN", raising=False)
        monkeypatch.setenv("MSGPU_GRAPH_THREADS", "1")
        assert L.msgpu_graph_max_span_tree(n, a.ctypes.data, b.ctypes.data, w.ctypes.data, cons.ctypes.data, m, want.ctypes.data) == 0
        monkeypatch.setenv("MSGPU_GRAPH_PAR_MIN", "1")
        monkeypatch.setenv("MSGPU_GRAPH_THREADS", "4")
        assert L.msgpu_graph_max_span_tree(n, a.ctypes.data, b.ctypes.data, w.ctypes.data, cons.ctypes.data, m, got.ctypes.data) == 0
        assert np.array_equal(got, want), case
        assert not got[cons == 2].any() and not got[a == b].any()


def test_connected_components_CC_test(L, oracle):
    fx = fixture("cc")
    p1, p2 = fx["phase1"], fx["phase2"]

    def product(vertices, edges):
        a, b = u32([e[0] for e in edges]), u32([e[1] for e in edges])
        n = max(vertices) + 1
        cons = np.ones(len(a), dtype=np.uint8)  # every edge e_POS
        comp, nc = np.zeros(n, dtype="<u4"), C.c_uint32()
        assert L.msgpu_graph_connected_components(n, a.ctypes.data, b.ctypes.data, cons.ctypes.data, len(a),
                                                  comp.ctypes.data, C.byref(nc)) == 0
        groups = {}
        for v in vertices:
            groups.setdefault(int(comp[v]), set()).add(v)
        return sorted(groups.values(), key=len, reverse=True)

    def restatement(vertices, edges):
        g, G = _ograph(vertices, edges)
        for x, y in edges:
            g.get_edge(x, y).set_consensus(True)
        return sorted((set(c) for c in G.connected_components(g)), key=len, reverse=True)

    for fn in (product, restatement):
        got = fn(p1["vertices"], p1["edges_pos"])
        assert len(got) == p1["n_components"] and got[0] == set(p1["component"])
        got = fn(p1["vertices"] + p2["added_vertices"], p1["edges_pos"] + p2["added_edges_pos"])
        assert len(got) == p2["n_components"]
        assert got[0] == set(p2["larger_component"]) and got[1] == set(p2["smaller_component"])


def test_shortest_path_Graph_test(L, oracle):
    fx = fixture("shortest_path")
    n = max(fx["vertices"]) + 1
    for directed, edges, want in ((0, fx["graph_edges"], fx["expect_undirected"]),
                                  (1, fx["digraph_edges"], fx["expect_directed"])):
        a, b = u32([e[0] for e in edges]), u32([e[1] for e in edges])
        path, k = np.zeros(n, dtype="<u4"), C.c_uint32(n)
        assert L.msgpu_graph_shortest_path(n, a.ctypes.data, b.ctypes.data, len(a), directed, fx["from"], fx["to"],
                                           path.ctypes.data, C.byref(k)) == 0
        assert [int(x) for x in path[:k.value]] == want
        g, G = _ograph(fx["vertices"], edges, directed=bool(directed))
        assert G.shortest_path(g, fx["from"], fx["to"]) == want


def test_topological_sort_Graph_test(L, oracle):
    fx = fixture("topological_sort")
    n = max(fx["vertices"]) + 1
    a, b = u32([e[0] for e in fx["digraph_edges"]]), u32([e[1] for e in fx["digraph_edges"]])
    order, k = np.zeros(n, dtype="<u4"), C.c_uint32()
    assert L.msgpu_graph_sort_topologically(n, a.ctypes.data, b.ctypes.data, len(a), order.ctypes.data, C.byref(k)) == 0
    got = [int(x) for x in order[:k.value] if int(x) in fx["vertices"]]  # id 0 is not a vertex of the test
    assert got == fx["expect_order"]
    g, G = _ograph(fx["vertices"], fx["digraph_edges"], directed=True)
    assert g.sort_topologically() == fx["expect_order"]


def test_readline_IO_test(L, oracle, tmp_path):
    fx = fixture("io_readline")
    path = os.path.join(HERE, fx["file"])
    data = open(path, "rb").read()
    n = C.c_size_t()
    assert L.msgpu_index_lines(path.encode(), None, 0, C.byref(n)) == 0
    assert n.value == fx["expect_n_lines"]
    off = np.zeros(n.value + 1, dtype="<u8")
    assert L.msgpu_index_lines(path.encode(), off.ctypes.data, len(off), C.byref(n)) == 0
    lines = [data[int(off[i]):int(off[i + 1])].decode() for i in range(n.value)]
    assert lines == fx["expect_lines"]  # the last line has no '\n' and is still a line
    # the loader counts the same lines; the oracles' loaders too (a PAF of that shape: 3 lines, the last unterminated)
    paf = tmp_path / "three.paf"
    row = "u%d\t1000\t0\t900\t+\tr%d\t5000\t10\t910\t800\t900\t60"
    paf.write_text("\n".join(row % (i, i) for i in range(3)))
    from muchsalsa_amd import overlap
    from oracle import ms_oracle_py as P
    p = overlap.parse_paf(str(paf))
    o = oracle.parse_paf(str(paf))
    rows_py, _, _ = P.parse_paf_text(paf.read_text())
    assert p.n_lines == o["n_lines"] == fx["expect_n_lines"]
    assert len(p.rows) == len(o["rows"]) == len(rows_py) == fx["expect_n_lines"] - 1  # BlastFileReader.cpp:76


def test_registry_Registry_test(L, oracle, tmp_path):
    fx = fixture("registry")
    r = L.msgpu_registry_new()
    for step in fx["steps"]:
        if step[0] == "clear":
            L.msgpu_registry_clear(r)
            assert L.msgpu_registry_size(r) == 0
        else:
            assert L.msgpu_registry_id(r, step[1].encode()) == step[2], step
    L.msgpu_registry_free(r)
    # the same numbering out of the loaders (product + both oracles): read names in the fixture's order, one file per
    # stretch between two clear()s
    from muchsalsa_amd import overlap
    from oracle import ms_oracle_py as P
    stretches, cur = [], []
    for step in fx["steps"]:
        if step[0] == "clear":
            stretches.append(cur)
            cur = []
        else:
            cur.append(step)
    stretches.append(cur)
    row = "u%d\t1000\t0\t900\t+\t%s\t5000\t10\t910\t800\t900\t60"
    for k, st in enumerate(stretches):
        paf = tmp_path / ("reg%d.paf" % k)
        paf.write_text("\n".join([row % (i, s[1]) for i, s in enumerate(st)] + [row % (99, "sentinel")]))
        p = overlap.parse_paf(str(paf))
        o = oracle.parse_paf(str(paf))
        rows_py, names_py, _ = P.parse_paf_text(paf.read_text())
        for i, s in enumerate(st):
            assert int(p.rows["read_id"][i]) == int(o["rows"]["read_id"][i]) == rows_py[i]["read_id"] == s[2]
            assert p.read_names[s[2]] == o["read_names"][s[2]] == names_py[s[2]] == s[1]


def test_toggle_Toggle_test(L):
    fx = fixture("toggle")
    val = dict(fx["constants"])
    # tShouldBeTrue = false; tShouldBeTrue *= false  ->  the product of toggles is XNOR (Toggle.h:127-153)
    t = fx["tShouldBeTrue"]
    val["tShouldBeTrue"] = bool(L.msgpu_toggle_mul(int(t["initial"]), int(t["times_equals"])))
    for row in fx["assertions"]:
        a = val[row["a"]]
        if row["op"] is None:
            got = a
        else:
            b = val[row["b"]]
            got = {"&&": a and b, "==": a == b, "!=": a != b}[row["op"]]
        if row["not"]:
            got = not got
        assert got == row["expect"], row
    for a in (0, 1):
        for b in (0, 1):
            assert L.msgpu_toggle_mul(a, b) == int(a == b)


GOP = dict(DELETE_EDGE=1, DELETE_VERTEX=2, ORDER=3, SIZE=4, HAS_EDGE=5, NEIGHBORS=6, PREDECESSORS=7, IN_DEGREE=8, OUT_DEGREE=9,
           SUBGRAPH=10, ARG=11)


def _bookkeeping(L, n, edges, directed, ops):
    """msgpu_graph_bookkeeping -> list of output words"""
    a, b = u32([e[0] for e in edges]), u32([e[1] for e in edges])
    o = np.ascontiguousarray(ops, dtype="<u4").reshape(-1, 3)
    out = np.zeros(64 + 8 * len(o) + 2 * len(edges), dtype="<u4")
    n_out = C.c_size_t()
    rc = L.msgpu_graph_bookkeeping(n, a.ctypes.data if len(a) else None, b.ctypes.data if len(b) else None, len(a), int(directed),
                                   o.ctypes.data if len(o) else None, len(o), out.ctypes.data, len(out), C.byref(n_out))
    assert rc == 0, rc
    return [int(x) for x in out[: n_out.value]]


@pytest.mark.parametrize("name", ["EdgeDeletionTest", "VertexDeletionTest", "NeighboorTest", "SubgraphTest", "DegreeTest"])
def test_graph_bookkeeping_Graph_test(L, name):
    """Graph_test.cpp:81-277, 333-391 -- deleteEdge, deleteVertex (with its cascade), getNeighbors / getPredecessors /
    getSuccessors, getSubgraph, getInDegrees / getOutDegrees, the double insertion of an edge -- replayed statement by
    statement through msgpu_graph_bookkeeping (the tombstone bookkeeping the clean-up's deletions and the component split
    run on) and through the oracle's Graph / DiGraph; every assertion of the reference's test is asserted on both."""
    from oracle import ms_graph_py as G
    fx = fixture("graph_bookkeeping")[name]
    assert fx["source"].startswith("libms/tests/Graph_test.cpp")
    n_checked = 0
    for gname, g in fx["graphs"].items():
        directed = g["directed"]
        if "subgraph_of" in g:
            parent = fx["graphs"][g["subgraph_of"]]
            p_vertices = [e["v"] for e in parent["events"] if e["op"] == "add_vertex"]
            p_edges = [(e["a"], e["b"]) for e in parent["events"] if e["op"] == "add_edge"]
            ids = {v: i for i, v in enumerate(p_vertices)}
            ops = [[GOP["SUBGRAPH"], 1, len(g["vertices"])]] + [[GOP["ARG"], ids[v], 0] for v in g["vertices"]]
            words = _bookkeeping(L, len(p_vertices), [(ids[a], ids[b]) for a, b in p_edges], directed, ops)
            order, size = words[0], words[1]
            sub_edges = [(p_vertices[words[2 + 2 * k]], p_vertices[words[3 + 2 * k]]) for k in range(size)]
            assert order == len(g["vertices"])
            vertices, edges = list(g["vertices"]), sub_edges
            og, _ = _ograph(p_vertices, p_edges, directed)
            osub = og.subgraph(g["vertices"]) if not directed else None  # (the oracle's DiGraph is built per component, no getSubgraph)
        else:
            vertices = [e["v"] for e in g["events"] if e["op"] == "add_vertex"]
            edges = [(e["a"], e["b"]) for e in g["events"] if e["op"] == "add_edge"]
            osub = None
        ids = {v: i for i, v in enumerate(vertices)}
        og, _ = (osub, None) if osub is not None else _ograph(vertices, edges, directed)
        ops, expect = [], []  # the script, and per query what the reference's test asserts about its output
        for e in g["events"]:
            op = e["op"]
            if op in ("add_vertex", "add_edge"):
                continue
            if op == "delete_edge":
                ops.append([GOP["DELETE_EDGE"], ids[e["a"]], ids[e["b"]]])
                expect.append(None)
                og.delete_edge(og.get_edge(e["a"], e["b"]))
            elif op == "delete_vertex":
                ops.append([GOP["DELETE_VERTEX"], ids[e["v"]], 0])
                expect.append(None)
                og.delete_vertex(e["v"])
            elif op == "expect_order":
                ops.append([GOP["ORDER"], 0, 0])
                expect.append(("word", e["value"]))
                assert og.order() == e["value"] or any(x["op"] == "add_vertex" for x in g["events"][g["events"].index(e):])
            elif op == "expect_size":
                ops.append([GOP["SIZE"], 0, 0])
                expect.append(("word", e["value"]))
            elif op == "expect_has_edge":
                ops.append([GOP["HAS_EDGE"], ids[e["a"]], ids[e["b"]]])
                expect.append(("word", 1 if e["value"] else 0))
                assert og.has_edge(e["a"], e["b"]) == e["value"]
            elif op == "expect_has_vertex":
                assert e["v"] in ids and og.has_vertex(e["v"])
                n_checked += 1
            elif op in ("expect_neighbors", "expect_successors", "expect_predecessors"):
                ops.append([GOP["PREDECESSORS" if op.endswith("predecessors") else "NEIGHBORS"], ids[e["v"]], 0])
                expect.append(("set", e["size"], sorted(e["ids"])))
                got = {"expect_neighbors": getattr(og, "neighbors", None), "expect_successors": og.successors,
                       "expect_predecessors": og.predecessors}[op](e["v"])
                got = [k for k, _ in got]  # (the oracle hands out (neighbour id, edge) pairs)
                assert sorted(got) == sorted(e["ids"]) and len(got) == e["size"]
            elif op in ("expect_in_degrees", "expect_out_degrees"):
                assert e["size"] == len(e["of"])
                for v, d in sorted(e["of"].items()):
                    ops.append([GOP["IN_DEGREE" if op == "expect_in_degrees" else "OUT_DEGREE"], ids[int(v)], 0])
                    expect.append(("word", d))
                    if osub is None and "subgraph_of" not in g:
                        assert len(og.predecessors(int(v)) if op == "expect_in_degrees" else og.successors(int(v))) == d
                # size of the degree map = living vertices
                ops.append([GOP["ORDER"], 0, 0])
                expect.append(("word", e["size"]))
            else:
                raise AssertionError(op)
        words = _bookkeeping(L, len(vertices), [(ids[a], ids[b]) for a, b in edges], directed, ops)
        at = 0
        for (o, x, y), ex in zip(ops, expect):
            if ex is None:
                continue
            if ex[0] == "word":
                # (an order asserted before the test has added its edges is the order after: adding edges adds no vertex)
                assert words[at] == ex[1], (gname, o, x, y, words[at], ex)
                at += 1
            else:
                cnt = words[at]
                got = [vertices[w] for w in words[at + 1: at + 1 + cnt]]
                assert cnt == ex[1] and got == ex[2], (gname, o, x, got, ex)
                at += 1 + cnt
            n_checked += 1
        assert at == len(words)
    assert n_checked >= 4


def test_graph_bookkeeping_rejects_bad_scripts(L):
    out = np.zeros(8, dtype="<u4")
    n_out = C.c_size_t()
    a, b = u32([0, 1]), u32([1, 2])
    bad = np.ascontiguousarray([[GOP["DELETE_VERTEX"], 7, 0]], dtype="<u4")
    assert L.msgpu_graph_bookkeeping(3, a.ctypes.data, b.ctypes.data, 2, 0, bad.ctypes.data, 1, out.ctypes.data, 8, C.byref(n_out)) != 0
    pred = np.ascontiguousarray([[GOP["PREDECESSORS"], 1, 0]], dtype="<u4")  # predecessors of an undirected graph
    assert L.msgpu_graph_bookkeeping(3, a.ctypes.data, b.ctypes.data, 2, 0, pred.ctypes.data, 1, out.ctypes.data, 8, C.byref(n_out)) != 0
    many = np.ascontiguousarray([[GOP["NEIGHBORS"], 1, 0]] * 4, dtype="<u4")  # 12 words into room for 8: told how many
    assert L.msgpu_graph_bookkeeping(3, a.ctypes.data, b.ctypes.data, 2, 0, many.ctypes.data, 4, out.ctypes.data, 8, C.byref(n_out)) != 0
    assert n_out.value == 12
