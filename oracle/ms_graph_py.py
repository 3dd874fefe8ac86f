"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the phases of MuCHSALSA between the overlap path and assemblePath:

  graph clean-up        src/main.cpp:194-288, 465-618   (contraction targets / roots, ContainElements, deletions,
                                                         computeBitweight, decycle)
  getMaxSpanTree        libms/src/kernel/mst.cpp:34-111  (Kruskal on edge weight, with the reference's union-find)
  getShortestPath       include/ms/graph/Graph.h:927-978
  getConnectedComponents libms/src/kernel/cc.cpp:33-70
  getDirectedGraph      libms/src/kernel/dg.cpp:35-121
  linearizeGraph        libms/src/kernel/lg.cpp:41-629   (sortReductionByWeight, findClusterWeights[Heuristic],
                                                         findConservationPathAlt, extractPaths, path joins)
  assemblePaths         src/main.cpp:620-661             (per connected component: directed graph -> paths)

Object model like the reference (Vertex / Edge / EdgeOrder objects shared between graphs, hash-map adjacency), small
inputs only.  PARITY UNPINNED: no reference test covers these functions and the reference cannot be built here.

Hash-order: the reference iterates unordered containers keyed by pointers (and one std::map / std::set ordered by
pointer VALUE, lg.cpp:419, main.cpp:211) in places where the order changes the result, so it differs between runs.
This restatement fixes one admissible order, the same the product uses ("canonical order"): vertices ascending id,
edges in creation order (the undirected graph's edges in (v1, v2) table order), neighbours ascending id, std::sort
ties stable, pointer-ordered containers ordered by vertex id.
"""
NONE, POS, NEG = 0, 1, -1
BASE_WEIGHT_MULTIPLICATOR = 1.1  # src/main.cpp:96
MAX_WEIGHT_MULTIPLICATOR = 0.8  # src/main.cpp:97


COVER = {}  # branch counters for the tests' coverage assertions


def _hit(name, n=1):
    COVER[name] = COVER.get(name, 0) + n


class GraphError(Exception):
    """The reference would terminate / hang / hit undefined behaviour here."""


class Vertex:
    def __init__(self, vid, length, meta0):
        self.id, self.length, self.meta0, self.direction = vid, length, meta0, NONE

    def set_direction(self, toggle):  # Vertex::setVertexDirection(bool), Vertex.h:202-204
        self.direction = POS if toggle else NEG


class Order:  # EdgeOrder, Edge.h:49-60
    def __init__(self, start, end, left, right, contained, base, score, ids, direction, primary):
        self.start, self.end, self.left, self.right, self.contained = start, end, left, right, contained
        self.base, self.score, self.ids, self.direction, self.primary = base, score, ids, direction, primary


class Edge:
    def __init__(self, a, b):
        self.v = (a, b)  # vertex ids (first, second)
        self.orders, self.shadow, self.weight, self.consensus = [], False, 0, NONE

    def set_consensus(self, toggle):
        self.consensus = POS if toggle else NEG


class GraphBase:
    bidirectional = False

    def __init__(self):
        self.vertices, self.adj, self.edges = {}, {}, {}  # id -> Vertex; id -> {id -> Edge}; id(edge) -> Edge

    def copy(self):  # GraphBase(GraphBase const &): shallow -- Vertex and Edge objects are shared, Graph.h:773-774
        g = type(self)()
        g.vertices, g.edges = dict(self.vertices), dict(self.edges)
        g.adj = {k: dict(v) for k, v in self.adj.items()}
        return g

    def add_vertex(self, v):
        self.vertices.setdefault(v.id, v)

    def has_vertex(self, vid):
        return vid in self.vertices

    def get_vertices(self):
        return [self.vertices[k] for k in sorted(self.vertices)]

    def get_edges(self):
        return list(self.edges.values())

    def _on_edge_added(self, a, b):
        pass

    def _on_edge_deleted(self, a, b):
        pass

    def _add_edge_internal(self, e):  # Graph.cpp:291-311
        a, b = e.v
        inner = self.adj.setdefault(a, {})
        inserted = b not in inner
        if inserted:
            inner[b] = e
        if self.bidirectional:
            self.adj.setdefault(b, {}).setdefault(a, e)
        if inserted:
            self.edges[id(e)] = e
        return inserted

    def add_edge(self, a, b):  # Graph.cpp:212-230
        if a not in self.vertices or b not in self.vertices:
            return
        e = Edge(a, b)
        if self._add_edge_internal(e):
            self._on_edge_added(a, b)

    def get_edge(self, a, b):
        return self.adj.get(a, {}).get(b)

    def has_edge(self, a, b):
        return b in self.adj.get(a, {})

    def delete_edge(self, e):  # Graph.cpp:232-258
        def erase(s, t):
            if s in self.adj and t in self.adj[s]:
                del self.adj[s][t]
                self._on_edge_deleted(s, t)
        erase(e.v[0], e.v[1])
        if self.bidirectional:
            erase(e.v[1], e.v[0])
        self.edges.pop(id(e), None)

    def successors(self, vid):  # ascending id (canonical)
        inner = self.adj.get(vid, {})
        return [(k, inner[k]) for k in sorted(inner)]

    def predecessors(self, vid):  # Graph.cpp:270-287
        return [(s, self.adj[s][vid]) for s in sorted(self.adj) if s != vid and vid in self.adj[s]]

    def delete_vertex(self, vid):  # Graph.cpp:151-209
        for t, e in list(self.adj.get(vid, {}).items()):
            if self.bidirectional and t in self.adj:
                self.adj[t].pop(vid, None)
            self.edges.pop(id(e), None)
            self._on_edge_deleted(e.v[0], e.v[1])
        self.adj.pop(vid, None)
        if not self.bidirectional:
            for s in list(self.adj):
                e = self.adj[s].pop(vid, None)
                if e is not None:
                    self.edges.pop(id(e), None)
                    self._on_edge_deleted(e.v[0], e.v[1])
        self.vertices.pop(vid, None)

    def order(self):
        return len(self.vertices)

    def size(self):
        return len(self.edges)


class Graph(GraphBase):
    bidirectional = True

    def neighbors(self, vid):
        return self.successors(vid)

    @staticmethod
    def from_parts(vertices, edges):  # GraphBase(vertices, edges, true), Graph.cpp:129-139
        g = Graph()
        g.vertices = dict(vertices)
        for e in edges:
            if e.v[0] in g.vertices and e.v[1] in g.vertices:
                g._add_edge_internal(e)
        return g

    def subgraph(self, vertex_ids):  # Graph::getSubgraph, Graph.cpp:317-326
        return Graph.from_parts({v: self.vertices[v] for v in vertex_ids}, self.get_edges())


class DiGraph(GraphBase):
    def __init__(self):
        super().__init__()
        self.indeg, self.outdeg = {}, {}

    def copy(self):
        g = super().copy()
        g.indeg, g.outdeg = dict(self.indeg), dict(self.outdeg)
        return g

    def add_vertex(self, v):  # Graph.cpp:332-342
        super().add_vertex(v)
        self.indeg.setdefault(v.id, 0)
        self.outdeg.setdefault(v.id, 0)

    def _on_edge_added(self, a, b):
        if a in self.outdeg:
            self.outdeg[a] += 1
        if b in self.indeg:
            self.indeg[b] += 1

    def _on_edge_deleted(self, a, b):
        if a in self.outdeg:
            self.outdeg[a] -= 1
        if b in self.indeg:
            self.indeg[b] -= 1

    def delete_vertex(self, vid):  # Graph.h:871-876
        super().delete_vertex(vid)
        self.indeg.pop(vid, None)
        self.outdeg.pop(vid, None)

    def sort_topologically(self):  # Graph.cpp:359-395
        nonnull = {v: d for v, d in self.indeg.items() if d > 0}
        ready = [v for v in sorted(self.indeg) if self.indeg[v] <= 0]
        result = []
        while ready:
            v = ready.pop()
            for t, _ in self.successors(v):
                nonnull[t] = nonnull.get(t, 0) - 1
                if nonnull[t] == 0:
                    ready.append(t)
                    del nonnull[t]
            result.append(v)
        return result


# ----------------------------------------------------------------------------------------------------------------------
# graph clean-up, src/main.cpp:180-288
# ----------------------------------------------------------------------------------------------------------------------

def build_graph(tables, read_len, read_first_line):
    """Graph + Edge/EdgeOrder objects from the flat result tables of the overlap path."""
    g = Graph()
    for vid in range(len(read_len)):
        g.add_vertex(Vertex(vid, int(read_len[vid]), int(read_first_line[vid])))
    edge_list, order_owner = [], {}
    for i, e in enumerate(tables["edges"]):
        a, b = int(e["v1"]), int(e["v2"])
        g.add_edge(a, b)
        ed = g.get_edge(a, b)
        ed.shadow = bool(e["shadow"])
        for k in range(int(e["order_off"]), int(e["order_off"]) + int(e["order_cnt"])):
            o = tables["orders"][k]
            fl = int(o["flags"])
            ids = [int(x) for x in tables["ids"][int(o["ids_off"]): int(o["ids_off"]) + int(o["ids_cnt"])]]
            od = Order(int(o["start"]), int(o["end"]), float(o["left_offset"]), float(o["right_offset"]), bool(fl & 2),
                       int(o["base"]), int(o["score"]), ids, bool(fl & 4), bool(fl & 8))
            ed.orders.append(od)
            order_owner[k] = od
        edge_list.append(ed)
    return g, edge_list, order_owner


def clean_up(g, edge_list, order_owner, contraction_order, vm_has):
    """src/main.cpp:194-288 after findContractionEdges.  contraction_order[e] = order-table index or -1;
    vm_has(read, anchor) = MatchMap::getVertexMatch(read, anchor) != nullptr.
    -> contain_elements {vertex id: [dict(nano, length, score, direction, primary, anchors)]}"""
    contraction = [(edge_list[e], order_owner[int(k)]) for e, k in enumerate(contraction_order) if k >= 0]
    targets = {v: v for v in g.vertices}  # :194-197
    for _, o in contraction:  # findContractionTargets, :465-482
        to = targets[o.end]
        if targets[o.start] == o.start or g.vertices[targets[o.start]].meta0 > g.vertices[to].meta0:
            targets[o.start] = to
    deletable, roots = set(), set()
    for _, o in contraction:  # findDeletableVertices, :484-507
        deletable.add(o.start)
        roots.add(targets[o.start])
        roots.discard(o.start)
    contain = {}
    for _, o in contraction:  # contract, :509-531
        if o.end not in roots:
            continue
        anchors = [a for a in o.ids if vm_has(o.start, a)]
        contain.setdefault(o.end, []).append(dict(nano=o.start, length=g.vertices[o.start].length, score=o.score,
                                                  direction=o.direction, primary=o.primary, anchors=anchors))
    for v in sorted(deletable):  # :242-244
        g.delete_vertex(v)
    dele = []
    for e in g.get_edges():  # findDeletableEdges, :534-549
        e.orders = [o for o in e.orders if not o.contained]
        if not e.orders:
            dele.append(e)
    for e in dele:
        g.delete_edge(e)
    edges = g.get_edges()  # :264
    for e in edges:  # computeBitweight, :551-573
        if not e.orders:
            continue
        if e.shadow:
            if all(o.direction == e.orders[0].direction for o in e.orders):
                e.set_consensus(e.orders[0].direction)
        else:
            e.weight = e.orders[0].score
            e.set_consensus(e.orders[0].direction)
    mst = max_span_tree(g)
    dele = {}
    for e in edges:  # decycle, :575-618
        if e.consensus != NONE and not mst.has_edge(e.v[0], e.v[1]):
            path = shortest_path(mst, e.v[0], e.v[1])
            if not path:
                raise GraphError("decycle: no tree path (std::prev of an empty vector)")
            direction = e.consensus == POS
            weights = []
            for a, b in zip(path, path[1:]):
                pe = g.get_edge(a, b)
                direction = direction == (pe.consensus == POS)
                weights.append(float(pe.weight))
            if not direction and weights:
                lo, hi = min(weights), max(weights)
                base = float(e.weight)
                if lo < base or (base * BASE_WEIGHT_MULTIPLICATOR >= lo and lo < hi * MAX_WEIGHT_MULTIPLICATOR):
                    i = weights.index(lo)
                    pe = g.get_edge(path[i], path[i + 1])
                    dele[id(pe)] = pe
                    _hit("decycle_weak_tree_edge")
                dele[id(e)] = e
                _hit("decycle")
    for e in dele.values():  # :285-287
        g.delete_edge(e)
    return contain


class _UnionFind:  # mst.cpp:35-73, including unify()'s use of the weights of the vertices rather than of their roots
    def __init__(self):
        self.parent, self.weight = {}, {}

    def find(self, v):
        if v not in self.parent:
            self.parent[v], self.weight[v] = v, 1
            return v
        path, root = [v], self.parent[v]
        while root != path[-1]:
            path.append(root)
            root = self.parent[root]
        for a in path:
            self.parent[a] = root
        return root

    def unify(self, v1, v2):
        first, second = self.find(v1), self.find(v2)
        if self.weight.setdefault(v2, 0) > self.weight.setdefault(v1, 0):
            first, second = second, first
        self.weight[first] = self.weight.setdefault(first, 0) + self.weight.setdefault(second, 0)
        self.parent[second] = first


def max_span_tree(g):  # mst.cpp:75-111
    edges = [e for e in g.get_edges() if e.consensus != NONE]
    edges.sort(key=lambda e: -e.weight)  # stable
    uf, result = _UnionFind(), []
    for e in edges:
        if uf.find(e.v[0]) != uf.find(e.v[1]):
            result.append(e)
            uf.unify(e.v[0], e.v[1])
    return Graph.from_parts(g.vertices, result)


def shortest_path(g, src, dst):  # GraphUtil::getShortestPath, Graph.h:927-978 (unit weights, insertion counter ties)
    import heapq
    paths, dist, seen = {src: [src]}, {}, {src: 0}
    heap, c = [(0, 0, src)], 1
    while heap:
        d, _, v = heapq.heappop(heap)
        if v in dist:
            continue
        dist[v] = d
        if v == dst:
            break
        # _getReachableVertices: Graph -> getNeighbors, DiGraph -> getSuccessors (Graph.h:982-992)
        for nb, _e in (g.neighbors(v) if isinstance(g, Graph) else g.successors(v)):
            nd = dist[v] + 1
            if nb not in dist and (nb not in seen or nd < seen[nb]):
                seen[nb] = nd
                heapq.heappush(heap, (nd, c, nb))
                paths[nb] = paths[v] + [nb]
                c += 1
    return paths.get(dst, [])


def connected_components(g):  # cc.cpp:33-70
    result, visited = [], set()
    for v in g.get_vertices():
        if v.id in visited:
            continue
        comp, queue = [v.id], [v.id]
        visited.add(v.id)
        while queue:
            cur = queue.pop(0)
            for nb, e in g.neighbors(cur):
                if nb not in visited and e.consensus != NONE:
                    comp.append(nb)
                    queue.append(nb)
                    visited.add(nb)
        result.append(comp)
    return result


# ----------------------------------------------------------------------------------------------------------------------
# getDirectedGraph, dg.cpp:35-121
# ----------------------------------------------------------------------------------------------------------------------

def get_directed_graph(graph, component, start):
    dg = DiGraph()
    stack = [(start, True)]
    while stack:
        cur, toggle = stack.pop()
        if not dg.has_vertex(cur):
            dg.add_vertex(graph.vertices[cur])
        if graph.vertices[cur].direction == NONE:
            dg.vertices[cur].set_direction(toggle)
        for nb, nedge in component.neighbors(cur):
            other = graph.vertices[nb]
            a, b = nedge.v
            other_exists = dg.has_vertex(nb)
            if other_exists:
                other_exists = other.direction != NONE
            if not other_exists:
                dg.add_vertex(component.vertices[nb])
            if dg.has_edge(a, b) or dg.has_edge(b, a):
                continue
            edge = graph.get_edge(a, b)
            for o in nedge.orders:
                flip = False
                if not o.direction and o.base == nb:
                    flip = not flip
                if not toggle:
                    flip = not flip
                s, t = (o.end, o.start) if flip else (o.start, o.end)
                ne = dg.get_edge(s, t)
                if ne is None:
                    if dg.has_edge(t, s):
                        _hit("two_way_edge")
                    dg.add_edge(s, t)
                    ne = dg.get_edge(s, t)
                    ne.shadow = edge.shadow
                    if not edge.shadow:
                        ne.weight = nedge.weight
                    ne.src = edge  # the EdgeMatches copied at dg.cpp:97-99 are those of the undirected edge
                ne.orders.append(o)
            if nedge.consensus == NONE:
                continue
            nxt = toggle == (nedge.consensus == POS)
            if not other_exists:
                stack.append((nb, nxt))
    return dg


# ----------------------------------------------------------------------------------------------------------------------
# linearizeGraph, lg.cpp
# ----------------------------------------------------------------------------------------------------------------------

def sort_reduction_by_weight(dg):  # lg.cpp:418-520
    nonnull = {v: d for v, d in dg.indeg.items() if d > 0}  # std::map ordered by pointer: canonical = by id
    null = [v for v in sorted(dg.indeg) if dg.indeg[v] <= 0]
    resolved, neighbors = set(), set()
    if nonnull:
        neighbors.add(min(nonnull))
    while True:
        while null:
            v = null.pop(0)
            resolved.add(v)
            for s, _e in dg.successors(v):
                nonnull[s] -= 1
                if nonnull[s] == 0:
                    null.append(s)
                    del nonnull[s]
                    neighbors.discard(s)
                else:
                    neighbors.add(s)
        if not nonnull:
            break
        min_edge, min_vertex, min_score = None, None, 0
        for cand in (sorted(nonnull) if not neighbors else sorted(neighbors)):
            for p, e in dg.predecessors(cand):
                if p not in resolved:
                    if min_edge is None or e.weight < min_score:
                        min_edge, min_vertex, min_score = e, cand, e.weight
        if min_edge is None:
            raise GraphError("sortReductionByWeight: null edge dereferenced")
        min_edge.shadow = True
        _hit("cycle_cut")
        dg.delete_edge(min_edge)
        nonnull[min_vertex] -= 1
        if nonnull[min_vertex] == 0:
            del nonnull[min_vertex]
            null.append(min_vertex)
            neighbors.discard(min_vertex)


def find_cluster_weights(dg):  # lg.cpp:144-264
    order = dg.sort_topologically()
    idx = {v: i for i, v in enumerate(order)}
    result = {id(e): 0 for e in dg.get_edges()}
    succ = {v: sorted(idx[t] for t, _ in dg.successors(v)) for v in order}
    pred = {v: sorted(idx[t] for t, _ in dg.predecessors(v)) for v in order}
    for v in order:
        cands = [(set(succ[v]), [idx[v]])]
        for i_out in succ[v]:
            active = order[i_out]
            for i_in in pred[active]:
                k = 0
                while k < len(cands):  # the loop bound grows with the emplace_back inside, :193
                    op, vis = cands[k]
                    if vis[-1] == i_in and i_out in op:
                        cands.append((op & set(succ[active]), vis + [i_out]))
                    k += 1
            filtered = []
            for a, ca in enumerate(cands):
                dominated = False
                for b, cb in enumerate(cands):
                    if a != b and ca[0] <= cb[0] and set(ca[1]) <= set(cb[1]):
                        dominated = True
                        break
                if not dominated:
                    filtered.append(ca)
            cands = filtered
        best, best_len = [], 0
        for _op, vis in cands:
            if len(vis) > best_len:
                best, best_len = [vis], len(vis)
            elif len(vis) == best_len:
                best.append(vis)
        for mv in best:
            c = len(mv) - 1
            for i in range(max(len(mv), 1) - 1):
                e = dg.get_edge(order[mv[i]], order[mv[i + 1]])
                result[id(e)] += c
                c -= 1
    return result


def find_cluster_weights_heuristic(dg):  # lg.cpp:72-141
    order = dg.sort_topologically()
    idx = {v: i for i, v in enumerate(order)}
    result = {id(e): 0 for e in dg.get_edges()}
    for v in order:
        cands = {v: [idx[v]]}
        for sid in sorted(idx[t] for t, _ in dg.successors(v)):
            w, best = order[sid], []
            for p, _e in dg.predecessors(w):
                if p in cands and len(cands[p]) > len(best):
                    best = cands[p]
            cands.setdefault(w, best + [idx[w]])
        # std::max_element over an unordered_map: the first longest in iteration order; canonical = ascending id
        best = None
        for k in sorted(cands):
            if best is None or len(cands[k]) > len(best):
                best = cands[k]
        c = len(best) - 1
        for i in range(max(len(best), 1) - 1):
            e = dg.get_edge(order[best[i]], order[best[i + 1]])
            result[id(e)] += c
            c -= 1
    return result


def find_conservation_path_alt(dg, cw):  # lg.cpp:267-344
    order = dg.sort_topologically()
    final, open_paths = [], {}
    for v in order:
        if dg.outdeg[v] == 0:
            if v not in open_paths:
                if not final:
                    final = [v]
            else:
                if len(open_paths[v][1]) > len(final):
                    final = open_paths[v][1]
                    open_paths[v] = (open_paths[v][0], [])
                else:
                    open_paths[v] = (open_paths[v][0], [])
            continue
        max_outs, max_out = [], 0
        for t, e in dg.successors(v):
            a, b = e.v
            if b != t:
                a, b = b, a
            w = cw[id(e)]
            if w > max_out:
                max_out, max_outs = w, [(a, b)]
            elif w == max_out:
                max_outs.append((a, b))
        for a, b in max_outs:
            if b in open_paths:
                # openPaths[pVertex] default-constructs an entry exactly when the C++ expression touches it (:320-321)
                if open_paths[b][0] < max_out:
                    take = True
                elif open_paths[b][0] == max_out:
                    take = len(open_paths[b][1]) < len(open_paths.setdefault(v, (0, []))[1]) + 1
                else:
                    take = False
                if take:
                    open_paths[b] = (max_out, open_paths.setdefault(v, (0, []))[1] + [b])
            else:
                if v in open_paths:
                    open_paths[b] = (max_out, open_paths[v][1] + [b])
                else:
                    open_paths[b] = (max_out, [a, b])
        open_paths[v] = (open_paths.get(v, (0, []))[0], [])
    return final


def extract_paths(dg):  # lg.cpp:347-414
    cyc = dg.copy()
    for e in cyc.get_edges():
        if e.shadow:
            cyc.delete_edge(e)
    sort_reduction_by_weight(cyc)
    cw = find_cluster_weights(cyc) if cyc.order() < 150000 else find_cluster_weights_heuristic(cyc)
    paths, visited = [], set()
    while cyc.size() > 0:
        longest = find_conservation_path_alt(cyc, cw)
        if not longest:
            raise GraphError("extractPaths: empty path (front() of an empty vector)")
        if len(longest) < 10:
            in_visit = any(p in visited for p, _ in dg.predecessors(longest[0]))
            out_visit = any(s in visited for s, _ in dg.successors(longest[-1]))
            if (not in_visit and not out_visit) or ((in_visit or out_visit) and len(longest) > 5):
                paths.append(longest)
            else:
                _hit("short_path_dropped")
        else:
            paths.append(longest)
        for v in longest:
            visited.add(v)
            cyc.delete_vertex(v)
    for v in cyc.get_vertices():
        paths.append([v.id])
    return paths


def linearize_graph(dg):  # lg.cpp:522-629
    paths = extract_paths(dg)
    color_corr = {i: i for i in range(len(paths))}
    color_len = {i: len(p) for i, p in enumerate(paths)}
    v2idx = {}
    for i, p in enumerate(paths):
        for v in p:
            v2idx.setdefault(v, i)
    joins = []
    for n, e in enumerate(dg.get_edges()):
        if e.shadow:
            a, b = e.v
            if a not in v2idx or b not in v2idx:
                continue
            i1, i2 = v2idx[a], v2idx[b]
            s1, s2 = paths[i1].index(a), paths[i2].index(b)
            l1_end, l2_end = color_len[i1] - s1 - 1, color_len[i2] - s2 - 1
            if i1 != i2 and l1_end < s1 and s2 < l2_end:
                joins.append((l1_end + s2, n, e))
    joins.sort(key=lambda j: j[0])  # std::sort on (distance, Edge*): canonical tie order = edge creation order
    for dist, _n, e in joins:
        if dist > 3:
            break
        a, b = e.v

        def color(i):
            while color_corr[i] != i:
                i = color_corr[i]
            return i
        c1, c2 = color(v2idx[a]), color(v2idx[b])
        if c1 == c2:
            continue
        if a not in paths[c1] or b not in paths[c2]:
            continue
        i1, i2 = paths[c1].index(a), paths[c2].index(b)
        if color_len[c1] - i1 - 1 + i2 != dist:
            continue
        _hit("join")
        paths[c1] = paths[c1][:i1 + 1] + paths[c2][i2:]
        paths[c2] = []
        color_corr[c2] = color_corr[c1]
        color_len[c1], color_len[c2] = len(paths[c1]), 0
    return [p for p in paths if len(p) > 1]


# ----------------------------------------------------------------------------------------------------------------------
# assemblePaths driver, src/main.cpp:300-310, 620-661
# ----------------------------------------------------------------------------------------------------------------------

def path_steps(dg, path, em_of):
    """What assemblePath reads from the directed graph for one path: per consecutive pair the EdgeOrders of
    diGraph.getEdge(a, b) and the EdgeMatch overlaps of that edge (em_of(v1, v2) -> {anchor: (lo, hi)})."""
    steps = []
    for a, b in zip(path, path[1:]):
        e = dg.get_edge(a, b)
        if e is None:
            raise GraphError("path edge missing in the directed graph (make_not_null(nullptr))")
        steps.append({"orders": [{"ids": list(o.ids), "score": o.score, "base": o.base} for o in e.orders],
                      "em": em_of(*e.src.v)})
    return steps


def assemble_all(g, em_of):
    """-> [(path as [{"id", "dir", "len"}], steps)] in assembly order (assemblyIdx = position + 0, main.cpp:300,673)"""
    out = []
    for comp in connected_components(g):
        sub = g.subgraph(comp)
        verts = sub.get_vertices()
        if not verts:
            continue
        start = verts[0]
        for v in verts:  # std::max_element: first of the longest
            if v.length > start.length:
                start = v
        dg = get_directed_graph(g, sub, start.id)
        for p in linearize_graph(dg):
            steps = path_steps(dg, p, em_of)
            out.append(([{"id": v, "dir": g.vertices[v].direction == POS, "len": g.vertices[v].length} for v in p],
                        steps))
    return out
