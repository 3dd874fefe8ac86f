"""The host graph stage (clean-up, span forest / decycle, components, getDirectedGraph, linearizeGraph: src/main.cpp:194-310,
465-661, dg.cpp, lg.cpp) at the size of BASELINE.json configs[1] -- 10 k reads -- against oracle/ms_graph_py.py, on the
shape the stage exists for: unitigs that tile the genome and reads of mixed length (synth.TILED), so that thousands of
contraction edges, hundreds of deleted vertices, dozens of components and paths of hundreds of reads all occur; and the
same tables with a tenth of the EdgeOrder directions flipped, which makes decycle() walk tree paths and leaves
components with several paths.  No GPU: the tables come from the C oracle here (tests/test_gpu_fullsize.py repeats it on
the tables and the contraction list the GPU produces, and at the size of configs[2])."""
import numpy as np
import pytest

from muchsalsa_amd import synth
from muchsalsa_amd.graph import GraphStage
from muchsalsa_amd.overlap import MsgpuError


def compare_stage(oracle, rows, t, co, threads=5):
    """GraphStage == ms_graph_py on (rows, tables, contraction list): alive sets, consensus directions, weights, vertex
    directions, ContainElements, paths and per-step orders / EdgeMatches.  -> the stage's counters"""
    from oracle.ms_graph_py import GraphError
    from test_graph_stage import oracle_stage
    try:
        state, contain, paths, dirs = oracle_stage(oracle, rows, t, co)
    except GraphError:
        g = GraphStage(t, t["read_len"], t["read_first_line"])
        with pytest.raises(MsgpuError):
            g.clean_up(co, rows)
            g.linearize(threads)
        return None
    g = GraphStage(t, t["read_len"], t["read_first_line"])
    g.clean_up(co, rows)
    got = g.state()
    for k in ("vertex_alive", "edge_alive"):
        assert np.array_equal(got[k], state[k]), k
    alive = state["edge_alive"]
    assert np.array_equal(got["edge_consensus"][alive], state["edge_consensus"][alive])
    assert np.array_equal(got["edge_weight"][alive], state["edge_weight"][alive])
    g.linearize(threads)
    st = g.stats
    assert st.n_paths == len(paths)
    got_contain = {}
    for i, (want_path, want_steps) in enumerate(paths):
        p, steps, cont = g.path(i)
        assert p == want_path, i
        assert steps == want_steps, i
        got_contain.update(cont)
    on_paths = {r["id"] for p, _ in paths for r in p}
    want_contain = {v: [dict(nano=c["nano"], dir=c["direction"], anchors=c["anchors"]) for c in lst]
                    for v, lst in contain.items() if v in on_paths}
    assert got_contain == want_contain
    va = state["vertex_alive"]
    assert np.array_equal(g.state()["vertex_direction"][va], dirs[va])
    assert st.n_contain_elements == sum(len(v) for v in contain.values())
    out = {k: int(getattr(st, k)) for k in ("n_contraction_edges", "n_deleted_vertices", "n_contain_elements",
                                           "n_decycled_edges", "n_vertices", "n_edges", "n_components", "n_paths",
                                           "n_path_reads")}
    out["longest_path"] = max((len(p) for p, _ in paths), default=0)
    out["multi_path_components"] = st.n_paths > st.n_components or len({p[0]["id"] for p, _ in paths}) < len(paths)
    g.close()
    return out


def flip_strands(t, seed, every=10):
    """the tables with EdgeOrder::direction flipped on one order in `every` and a quarter of the shadow flags toggled:
    odd cycles for decycle(), edges without a consensus direction, components that fall apart into several paths"""
    rng = np.random.default_rng(seed)
    t2 = dict(t, edges=t["edges"].copy(), orders=t["orders"].copy())
    t2["orders"]["flags"] ^= np.where(rng.integers(0, every, len(t2["orders"])) == 0, 4, 0).astype(np.uint32)
    t2["edges"]["shadow"] ^= (rng.integers(0, 4, len(t2["edges"])) == 0).astype(np.uint8)
    return t2


@pytest.mark.parametrize("par_min", [None, "512"])
def test_tiled_cfg2_graph_stage_matches_restatement(oracle, monkeypatch, par_min):
    """par_min = 512: the loops that go to all host threads on graphs of 65 k items and more (adjacency build, per-edge
    clean-up loops, decycle, the directed edges of getDirectedGraph) do so here too, with grains that cut this graph."""
    if par_min:
        monkeypatch.setenv("MSGPU_GRAPH_PAR_MIN", par_min)
        monkeypatch.setenv("MSGPU_GRAPH_THREADS", "5")
    rows = synth.synth_rows(**synth.TILED["cfg2"])
    t = oracle.overlap(rows)
    n = len(t["read_len"])
    assert n == 10_000 and len(t["edges"]) > 50_000
    c = compare_stage(oracle, rows, t, oracle.find_contraction_edges(t, n))
    assert c["n_contraction_edges"] > 5_000 and c["n_deleted_vertices"] > 3_000 and c["n_contain_elements"] > 3_000
    assert c["n_components"] > 20 and c["n_paths"] > 20 and c["longest_path"] > 100
    t2 = flip_strands(t, 7)
    c2 = compare_stage(oracle, rows, t2, oracle.find_contraction_edges(t2, n))
    assert c2 is not None and c2["n_decycled_edges"] > 0 and c2["n_paths"] > c2["n_components"]
