#!/usr/bin/env python3
"""Known-answer vectors held by the reference's OWN unit tests, extracted as DATA into tests/golden/ref_tests/*.json.

The reference's tests (libms/tests/*.cpp, googletest) pin a handful of things the hot path relies on: the max span tree,
connected components, shortest path, topological order, line splitting, Registry numbering, the Toggle truth table.
This script reads those test files as text and pulls out NUMBERS and STRING LITERALS only (vertex ids, edge pairs,
weights, expected values) with regular expressions -- no line of the reference's source is copied.  Each fixture names
the file:line range it came from.  Run in the build container (needs /root/reference); the fixtures are committed, so
the tests (tests/test_ref_test_vectors.py) need no reference.

    python tools/make_ref_test_fixtures.py [/root/reference]
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_tests")

PAIR = r"std::make_pair\(\s*spVertex(\d+)\.get\(\),\s*spVertex(\d+)\.get\(\)\s*\)"


def test_body(text, suite, name):
    """(body, first line, last line) of TEST(suite, name) { ... }"""
    m = re.search(r"TEST\(%s,\s*%s\)\s*\{" % (suite, name), text)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1], text.count("\n", 0, m.start()) + 1, text.count("\n", 0, i) + 1


def pairs(body, prefix):
    return [[int(a), int(b)] for a, b in re.findall(prefix + PAIR, body)]


def mst(ref):
    body, l0, l1 = test_body(open(os.path.join(ref, "libms/tests/MST_test.cpp")).read(), "MSTTest", "BasicTest")
    vertices = [int(v) for v in re.findall(r"Vertex>\((\d+),\s*\d+\)", body)]
    edges = pairs(body, r"graph\.addEdge\(")
    weight = {(int(a), int(b)): int(w) for a, b, w in re.findall(r"getEdge\(" + PAIR + r"\)->setWeight\((\d+)\)", body)}
    with_consensus = [[int(a), int(b)] for a, b in re.findall(r"getEdge\(" + PAIR + r"\)->setConsensusDirection\(true\)", body)]
    sizes = [int(x) for x in re.findall(r"ASSERT_EQ\(mst\.getSize\(\),\s*(\d+)\)", body)]
    return {"source": "libms/tests/MST_test.cpp:%d-%d" % (l0, l1), "vertices": vertices,
            "edges": [[a, b, weight[(a, b)]] for a, b in edges], "consensus_true": with_consensus,
            "expect": {"size_before_consensus": sizes[0], "size": sizes[1], "order_equals_graph": True,
                       "has_edge": pairs(body, r"ASSERT_TRUE\(mst\.hasEdge\("),
                       "has_no_edge": pairs(body, r"ASSERT_FALSE\(mst\.hasEdge\(")}}


def cc(ref):
    body, l0, l1 = test_body(open(os.path.join(ref, "libms/tests/CC_test.cpp")).read(), "CCTest", "BasicTest")
    cut = body.index("spVertex6")  # the second half adds vertices 6..8
    first, second = body[:cut], body[cut:]
    sizes = [int(x) for x in re.findall(r"ASSERT_EQ\(cc\.size\(\),\s*(\d+)\)", body)]

    def members(part):
        out = {}
        for idx, v in re.findall(r"contains\(cc\[(\w+)\],\s*spVertex(\d+)\.get\(\)\)", part):
            out.setdefault(idx, []).append(int(v))
        return out
    return {"source": "libms/tests/CC_test.cpp:%d-%d" % (l0, l1),
            "phase1": {"vertices": [int(v) for v in re.findall(r"Vertex>\((\d+),\s*\d+\)", first)],
                       "edges_pos": pairs(first, r"graph\.addEdge\("), "n_components": sizes[0],
                       "component": members(first)["0"]},
            "phase2": {"added_vertices": [int(v) for v in re.findall(r"Vertex>\((\d+),\s*\d+\)", second)],
                       "added_edges_pos": pairs(second, r"graph\.addEdge\("), "n_components": sizes[1],
                       "larger_component": members(second)["firstIdx"],
                       "smaller_component": members(second)["secondIdx"]}}


def graph(ref):
    text = open(os.path.join(ref, "libms/tests/Graph_test.cpp")).read()
    body, l0, l1 = test_body(text, "GraphTest", "ShortestPathTest")
    v1, v2 = (int(x) for x in re.search(r"v1\s*=\s*(\d+),\s*v2\s*=\s*(\d+)", body).groups())
    und, dire = body.split("shortestPathVertices =")[0:2] + [""], None
    ids = [int(x) for x in re.findall(r"shortestPath\[\d+\]->getId\(\),\s*(\d+)\)", body)]
    lens = [int(x) for x in re.findall(r"ASSERT_EQ\(shortestPath\.size\(\),\s*(\d+)\)", body)]
    sp = {"source": "libms/tests/Graph_test.cpp:%d-%d" % (l0, l1),
          "vertices": sorted({int(v) for v in re.findall(r"Vertex>\((\d+),\s*\d+\)", body)}),
          "graph_edges": pairs(body, r"\bgraph\.addEdge\("), "digraph_edges": pairs(body, r"\bdiGraph\.addEdge\("),
          "from": v1, "to": v2, "expect_undirected": ids[:lens[0]], "expect_directed": ids[lens[0]:lens[0] + lens[1]]}
    body, l0, l1 = test_body(text, "GraphTest", "TopologicalSortTest")
    order = [int(v) for _, v in sorted((int(i), v) for i, v in re.findall(r"sortedGraph\[(\d+)\],\s*spVertex(\d+)\.get\(\)", body))]
    topo = {"source": "libms/tests/Graph_test.cpp:%d-%d" % (l0, l1),
            "vertices": [int(v) for v in re.findall(r"Vertex>\((\d+),\s*\d+\)", body)],
            "digraph_edges": pairs(body, r"diGraph\.addEdge\("), "expect_order": order}
    return sp, topo


def bookkeeping(ref):
    """EdgeDeletion / VertexDeletion / Neighboor / Subgraph / Degree (Graph_test.cpp:81-277, 333-391) as event lists: per graph
    object of a test, in statement order -- add_edge, delete_edge, delete_vertex, and what the test asserts afterwards (order,
    size, has_edge, neighbour sets, degrees, a subgraph and the assertions on it).  Numbers and object names only."""
    text = open(os.path.join(ref, "libms/tests/Graph_test.cpp")).read()
    out = {}
    for name in ("EdgeDeletionTest", "VertexDeletionTest", "NeighboorTest", "SubgraphTest", "DegreeTest"):
        body, l0, l1 = test_body(text, "GraphTest", name)
        alias, named_edge, ptr_edge, sets, degs, vec = {}, {}, {}, {}, {}, {}
        graphs = {}

        def vtx(expr):
            expr = expr.strip()
            m = re.fullmatch(r"(?:gsl::make_not_null\()?spVertex(\d+)(?:\.get\(\)|->getSharedPtr\(\))\)?", expr)
            if m:
                return int(m.group(1))
            m = re.fullmatch(r"(\w+)->getId\(\)", expr)
            if m:
                return alias[m.group(1)]
            m = re.fullmatch(r"(\w+)\.(first|second)", expr)
            if m:
                return named_edge[m.group(1)][0 if m.group(2) == "first" else 1]
            return alias[expr]

        def pair(expr):
            expr = expr.strip()
            if expr in named_edge:
                return list(named_edge[expr])
            m = re.fullmatch(r"std::make_pair\((.+?),\s*(.+)\)", expr)
            return [vtx(m.group(1)), vtx(m.group(2))]

        def ev(g, **kw):
            graphs.setdefault(g, {"events": []})["events"].append(kw)

        for st in (x.strip() for x in re.sub(r"//[^\n]*", "", body).split(";")):
            st = " ".join(st.split())
            if not st:
                continue
            m = re.fullmatch(r"auto (\w+) = muchsalsa::graph::(Graph|DiGraph)\(\)", st)
            if m:
                graphs[m.group(1)] = {"directed": m.group(2) == "DiGraph", "events": []}
                continue
            m = re.fullmatch(r"auto (\w+Edge) = std::make_pair\((.+?), (.+)\)", st)
            if m:
                named_edge[m.group(1)] = (vtx(m.group(2)), vtx(m.group(3)))
                continue
            m = re.fullmatch(r"auto (\w+) = (spVertex\d+\.get\(\)), (\w+) = (spVertex\d+\.get\(\))", st)
            if m:
                alias[m.group(1)], alias[m.group(3)] = vtx(m.group(2)), vtx(m.group(4))
                continue
            m = re.fullmatch(r"auto const (\w+) = std::vector<[^>]*>\(\{(.+)\}\)", st)
            if m:
                vec[m.group(1)] = [vtx(x) for x in m.group(2).split(",")]
                continue
            m = re.fullmatch(r"(\w+)\.addVertex\((.+)\)", st)
            if m:
                ev(m.group(1), op="add_vertex", v=vtx(re.sub(r"std::move\((\w+)\)", r"\1.get()", m.group(2))))
                continue
            m = re.fullmatch(r"(\w+)\.addEdge\((.+)\)", st)
            if m:
                a, b = pair(m.group(2))
                ev(m.group(1), op="add_edge", a=a, b=b)
                continue
            m = re.fullmatch(r"(?:auto \*)?(\w+) = (\w+)\.getEdge\((.+)\)", st)
            if m:
                ptr_edge[m.group(1)] = (m.group(2), pair(m.group(3)))
                continue
            m = re.fullmatch(r"(\w+)\.deleteEdge\((\w+)\)", st)
            if m:
                a, b = ptr_edge[m.group(2)][1]
                ev(m.group(1), op="delete_edge", a=a, b=b)
                continue
            m = re.fullmatch(r"(\w+)\.deleteVertex\((.+)\)", st)
            if m:
                ev(m.group(1), op="delete_vertex", v=vtx(m.group(2)))
                continue
            m = re.fullmatch(r"ASSERT_EQ\((\w+)\.get(Order|Size)\(\), (\d+)\)", st)
            if m:
                ev(m.group(1), op="expect_" + m.group(2).lower(), value=int(m.group(3)))
                continue
            m = re.fullmatch(r"ASSERT_(TRUE|FALSE)\((\w+)\.hasEdge\((.+)\)\)", st)
            if m:
                a, b = pair(m.group(3))
                ev(m.group(2), op="expect_has_edge", a=a, b=b, value=m.group(1) == "TRUE")
                continue
            m = re.fullmatch(r"ASSERT_TRUE\((\w+)\.hasVertex\((.+)\)\)", st)
            if m:
                ev(m.group(1), op="expect_has_vertex", v=vtx(m.group(2)))
                continue
            m = re.fullmatch(r"ASSERT_NE\((\w+), nullptr\)", st)
            if m:
                g, (a, b) = ptr_edge[m.group(1)]
                ev(g, op="expect_has_edge", a=a, b=b, value=True)
                continue
            m = re.fullmatch(r"(?:auto(?: const)? )?(\w+) = (\w+)\.get(Neighbors|Predecessors|Successors)\((.+)\)", st)
            if m:
                e = {"op": "expect_" + m.group(3).lower(), "v": vtx(m.group(4)), "ids": [], "size": None}
                graphs[m.group(2)]["events"].append(e)
                sets[m.group(1)] = e
                continue
            m = re.fullmatch(r"(?:auto(?: const)? &?)?(\w+) = (\w+)\.get(In|Out)Degrees\(\)", st)
            if m:
                e = {"op": "expect_%s_degrees" % m.group(3).lower(), "of": {}, "size": None}
                graphs[m.group(2)]["events"].append(e)
                degs[m.group(1)] = e
                continue
            m = re.fullmatch(r"auto const (\w+) = (\w+)\.getSubgraph\((\w+)\)", st)
            if m:
                graphs[m.group(1)] = {"directed": graphs[m.group(2)]["directed"], "subgraph_of": m.group(2),
                                      "vertices": vec[m.group(3)], "events": []}
                continue
            m = re.fullmatch(r"ASSERT_EQ\((\w+)\.size\(\), (\d+)\)", st)
            if m:
                (sets.get(m.group(1)) or degs[m.group(1)])["size"] = int(m.group(2))
                continue
            m = re.fullmatch(r"ASSERT_TRUE\((\w+)\.contains\((\d+)\)\)", st)
            if m:
                sets[m.group(1)]["ids"].append(int(m.group(2)))
                continue
            m = re.fullmatch(r"ASSERT_EQ\((\w+)\.at\((.+)\), (\d+)\)", st)
            if m:
                degs[m.group(1)]["of"][str(vtx(m.group(2)))] = int(m.group(3))
                continue
            if re.match(r"auto spVertex\d+ = std::make_shared", st):
                continue
            raise SystemExit("Graph_test.cpp %s: statement not understood: %r" % (name, st))
        out[name] = {"source": "libms/tests/Graph_test.cpp:%d-%d" % (l0, l1), "graphs": graphs}
    return out


def io(ref):
    body, l0, l1 = test_body(open(os.path.join(ref, "libms/tests/IO_test.cpp")).read(), "IOTest", "ReadlineTest")
    lines = [s.encode().decode("unicode_escape") for s in re.findall(r'ASSERT_EQ\(result\[\d+\],\s*"((?:[^"\\]|\\.)*)"\)', body)]
    n = int(re.search(r"ASSERT_EQ\(result\.size\(\),\s*(\d+)\)", body).group(1))
    shutil.copyfile(os.path.join(ref, "test_data", "text.txt"), os.path.join(OUT, "text.txt"))  # a data file
    return {"source": "libms/tests/IO_test.cpp:%d-%d + test_data/text.txt" % (l0, l1), "file": "text.txt",
            "expect_n_lines": n, "expect_lines": lines}


def registry(ref):
    body, l0, l1 = test_body(open(os.path.join(ref, "libms/tests/Registry_test.cpp")).read(), "RegistryTest", "Test")
    steps = []
    for m in re.finditer(r'registry\["([^"]*)"\],\s*(\d+)\)|registry\.(clear)\(\)', body):
        steps.append(["clear"] if m.group(3) else ["lookup", m.group(1), int(m.group(2))])
    return {"source": "libms/tests/Registry_test.cpp:%d-%d" % (l0, l1), "steps": steps}


def toggle(ref):
    body, l0, l1 = test_body(open(os.path.join(ref, "libms/tests/Toggle_test.cpp")).read(), "ToggleTest", "BasicTest")
    val = {"tTrue": True, "tFalse": False}
    init = re.search(r"tShouldBeTrue\s*=\s*(true|false)", body).group(1) == "true"
    mul = re.search(r"tShouldBeTrue\s*\*=\s*(true|false)", body).group(1) == "true"
    rows = []
    for expr, want in re.findall(r"ASSERT_EQ\(\(\(bool\)\(?(.+?)\)?\),\s*(true|false)\)", body):
        expr = expr.strip("() ")
        m = re.fullmatch(r"(!?)(\w+)(?:\s*(&&|==|!=)\s*(\w+))?", expr)
        rows.append({"not": bool(m.group(1)), "a": m.group(2), "op": m.group(3), "b": m.group(4), "expect": want == "true"})
    return {"source": "libms/tests/Toggle_test.cpp:%d-%d" % (l0, l1), "constants": val,
            "tShouldBeTrue": {"initial": init, "times_equals": mul}, "assertions": rows}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    os.makedirs(OUT, exist_ok=True)
    sp, topo = graph(ref)
    for name, data in (("mst", mst(ref)), ("cc", cc(ref)), ("shortest_path", sp), ("topological_sort", topo),
                       ("graph_bookkeeping", bookkeeping(ref)),
                       ("io_readline", io(ref)), ("registry", registry(ref)), ("toggle", toggle(ref))):
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
            f.write("\n")
        print(name, json.dumps(data)[:160])


if __name__ == "__main__":
    main()
