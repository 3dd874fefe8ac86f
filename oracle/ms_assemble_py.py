"""TEST INFRASTRUCTURE ONLY -- CPU restatement of muchsalsa::assemblePath (libms/src/kernel/ap.cpp:615-1362).

String-based and dictionary-based like the reference; small cases only (pure-Python loops).  Parity status: *unpinned*
(the reference has no test for this function and cannot be built here); it follows ap.cpp line by line and
each block cites the lines it restates.  Segment builders come from ms_oracle_py (ap.cpp:352-579).

Where the reference iterates a hash container keyed by pointers or ids (std::unordered_map / unordered_set), its order
is unspecified and differs between runs and standard libraries.  This restatement fixes ONE admissible order and the
product uses the same one (DESIGN.md "canonical orders"):
  * vertices of a graph: ascending vertex id;   * edges of the anchor DAG: creation order;
  * successors / predecessors of a vertex: ascending neighbour id;   * entries of a tap: ascending anchor-vertex id;
  * std::sort on elements the comparator calls equivalent: stable.

Inputs (plain Python):
  path       [{"id": read id, "dir": True for Direction::e_POS / False for e_NEG / None for e_NONE (a vertex
             getDirectedGraph never reached through an edge with a consensus direction), "len": getNanoporeLength()}, ...]
  steps      one per consecutive pair of the path: {"orders": [{"ids": [...], "score": int, "base": read id}, ...],
             "em": {anchor id: (ov_lo, ov_hi)}}    -- EdgeOrders of diGraph.getEdge(path[i], path[i+1]) + its EdgeMatches
  vm         {(read id, anchor id): row}           -- MatchMap::getVertexMatch (row = record with i_lo,i_hi,n_lo,n_hi,flags)
  contains   {read id: [{"nano": id, "dir": bool, "matches": {anchor id: row}}, ...]}   -- ContainElement lists
  nano, illu {id: bytes}
"""
import functools

from .ms_oracle_py import (_corrected_range, _vm, get_anchor_sequence, get_sequence, get_sequence_between_anchors,
                           get_sequence_left_of_anchor, get_sequence_right_of_anchor, str_slice)

SEQUENCE_LINE_LENGTH = 60  # ap.cpp:52
TH_SEQUENCE_LENGTH = 200  # ap.cpp:53


class AssemblyError(Exception):
    """The reference would terminate / hang / hit undefined behaviour here."""


def limit_length(s):  # ap.cpp:61-76
    return b"\n".join(s[i:i + SEQUENCE_LINE_LENGTH] for i in range(0, len(s), SEQUENCE_LINE_LENGTH))


def tuple_tuple2id(t):  # ap.cpp:78-89
    return "%d,%d,%d" % (t[0][0], t[0][1], t[1])


def ramsey_r2(adj, vertices):  # ap.cpp:91-115
    if not vertices:
        return []
    first = vertices[0]
    neighbors = [v for v in vertices[1:] if v in adj[first]]
    non_neighbors = [v for v in vertices[1:] if v not in adj[first]]
    cn = ramsey_r2(adj, neighbors)
    cnn = ramsey_r2(adj, non_neighbors)
    cn.append(first)
    return cn if len(cn) >= len(cnn) else cnn


def get_anchor_cliques(adj):  # ap.cpp:117-138
    vertices = set(adj)
    current = ramsey_r2(adj, sorted(vertices))
    cliques = [current]
    while vertices:
        vertices -= set(current)
        current = ramsey_r2(adj, sorted(vertices))
        if current:
            cliques.append(current)
    return cliques


def get_cluster_anchors(cluster_modifier, id2overlap, steps, illumina_id, edge_idx):  # ap.cpp:140-189
    adj = {}
    for e1 in edge_idx:
        adj.setdefault(e1, set())
        for e2 in edge_idx:
            if e1 == e2:
                break
            o1, o2 = steps[e1]["em"][illumina_id], steps[e2]["em"][illumina_id]
            if max(o1[0], o2[0]) <= min(o1[1], o2[1]):
                adj[e2].add(e1)
                adj[e1].add(e2)
    for idx, clique in enumerate(get_anchor_cliques(adj)):
        common = None
        for v in clique:
            cluster_modifier[v][illumina_id] = idx
            ov = steps[v]["em"][illumina_id]
            common = ov if common is None else (max(common[0], ov[0]), min(common[1], ov[1]))
        id2overlap[(illumina_id, idx)] = (int(common[0]), int(common[1]))


class _Base:  # the (sequence, borderLeft, borderRight) triple of updateConsensusBase, ap.cpp:205-229
    def __init__(self):
        self.seq, self.lo, self.hi = None, 0, 0

    def update(self, new, lo, hi):
        if self.seq is None:
            self.seq, self.lo, self.hi = new, lo, hi
            return
        if lo < self.lo:
            self.seq = str_slice(new, 0, self.lo - lo) + self.seq
        elif hi > self.hi:
            self.seq = self.seq + str_slice(new, -(hi - self.hi), len(new))
        self.lo, self.hi = min(self.lo, lo), max(self.hi, hi)

    def copy(self):
        c = _Base()
        c.seq, c.lo, c.hi = self.seq, self.lo, self.hi
        return c


class _Adg:  # the anchor DiGraph of one path
    def __init__(self):
        self.succ, self.pred, self.edges, self.edge_of = {}, {}, [], {}

    def add_vertex(self, v):
        self.succ[v], self.pred[v] = {}, {}

    def add_edge(self, u, v):  # GraphBase::_addEdgeInternal: an existing edge is kept, Graph.cpp:291-311
        if v not in self.succ[u]:
            self.edge_of[(u, v)] = len(self.edges)
            self.edges.append((u, v))
            self.succ[u][v] = self.pred[v][u] = self.edge_of[(u, v)]
        return self.edge_of[(u, v)]

    def vertices(self):
        return sorted(self.succ)

    def sort_topologically(self):  # DiGraph::sortTopologically, Graph.cpp:359-395
        indeg = {v: len(self.pred[v]) for v in self.vertices()}
        ready = [v for v in self.vertices() if indeg[v] == 0]
        result = []
        while ready:
            v = ready.pop()
            for t in sorted(self.succ[v]):
                indeg[t] -= 1
                if indeg[t] == 0:
                    ready.append(t)
            result.append(v)
        return result


def _visit_ordered(visited, tap, adg, reg2id, pos_of, order, distances, sequences, anchor_seq, id2overlap, start):
    # ap.cpp:231-349
    base = _Base()
    queue_edges = set()  # ordered by (first ascending, second descending), :244-250
    queue_vertices = {pos_of[start]}
    while queue_vertices:
        idx = min(queue_vertices)
        queue_vertices.discard(idx)
        v = order[idx]

        def first_edge():
            return min(queue_edges, key=lambda e: (e[0], -e[1])) if queue_edges else None

        if not visited.get(v, False):
            visited[v] = True
            for t in sorted(adg.succ[v]):
                queue_edges.add((pos_of[t], idx))
                queue_vertices.add(pos_of[t])
            while queue_edges and first_edge()[0] == idx:
                e = first_edge()
                left, right = order[e[1]], order[e[0]]
                has_l, has_r = left in tap, right in tap
                ov_l, ov_r = id2overlap[reg2id[left]], id2overlap[reg2id[right]]
                edge = adg.edge_of[(left, right)]
                offset = distances[edge]
                len_l, len_r = ov_l[1] - ov_l[0] + 1, ov_r[1] - ov_r[0] + 1
                if has_l and not has_r:  # :295-307
                    pos = tap[left][1]
                    tap[right] = (pos + offset + 1, pos + offset + len_r)
                    if offset > 0:
                        base.update(sequences[edge][0], pos + 1, pos + offset)
                    base.update(anchor_seq[right], *tap[right])
                elif not has_l and has_r:  # :308-320
                    pos = tap[right][0]
                    tap[left] = (pos - offset - len_l, pos - offset - 1)
                    if offset > 0:
                        base.update(sequences[edge][0], pos - offset, pos)
                    base.update(anchor_seq[left], *tap[left])
                elif not has_l and not has_r:  # :321-337
                    tap[left] = (0, len_l - 1)
                    tap[right] = (len_l + offset, len_l + offset + len_r - 1)
                    if offset > 0:
                        base.update(sequences[edge][0], len_l, len_l + offset - 1)
                    base.update(anchor_seq[left], *tap[left])
                    base.update(anchor_seq[right], *tap[right])
                queue_edges.discard(e)
        else:
            while queue_edges and first_edge()[0] == idx:
                queue_edges.discard(first_edge())
    return base


def assemble_path(path, steps, vm, contains, nano, illu, asm_idx):
    """-> dict(target=bytes, target_fa=bytes, query_fa=bytes, paf=bytes, queries=[(name, bytes, lb, rb)],
               id2overlap={(anchor, clique): (lo, hi)}, tap={adg vertex: (lo, hi)}, left_most=int)"""
    if len(path) < 2:
        raise AssemblyError("path with fewer than two reads")
    cover = dict(multi_order=0, kinks=0, multi_clique=0, flips=0, nr_ties=0, extra_groups=0, dup_edges=0, no_seq=0,
                 contain_records=0)  # which branches this input exercised (for the tests' coverage assertions)
    is_neg = {p["id"]: p["dir"] is not None and not p["dir"] for p in path}  # getVertexDirection() == e_NEG

    # ---- candidate selection over the EdgeOrders of the path edges, ap.cpp:621-706 --------------------------------
    def find_best(cands):  # :633-642
        min_kinks = max_score = None
        for c in cands:
            if min_kinks is None or c["kinks"] < min_kinks or (c["kinks"] == min_kinks and c["score"] > max_score):
                min_kinks, max_score = c["kinks"], c["score"]
        return min_kinks, max_score

    candidates = [dict(open=set(), visited=set(), score=0, kinks=0, edges=[], orders=[], modifiers=[])]
    for i in range(len(path) - 1):
        nxt = []
        for order in steps[i]["orders"]:
            sub = []
            for c in candidates:
                ids = [int(x) for x in order["ids"]]
                if is_neg[order["base"]]:
                    ids.reverse()
                mods = [x for x in ids if x not in c["open"] and x in c["visited"]]
                sub.append(dict(open=set(ids), visited=c["visited"] | set(ids), score=c["score"] + int(order["score"]),
                                kinks=c["kinks"] + len(mods), edges=c["edges"] + [i], orders=c["orders"] + [order],
                                modifiers=c["modifiers"] + [mods]))
            mk, ms = find_best(sub)
            nxt += [c for c in sub if c["kinks"] == mk and c["score"] == ms]
        candidates = nxt
        cover["multi_order"] += len(steps[i]["orders"]) > 1
    if not candidates:
        raise AssemblyError("a path edge has no EdgeOrder (the reference dereferences end())")
    mk, ms = find_best(candidates)
    best = next(c for c in candidates if c["kinks"] == mk and c["score"] == ms)
    n_edges = len(best["edges"])
    cover["kinks"] = best["kinks"]

    # ---- anchor clusters -> cliques -> common overlaps, :708-719 ---------------------------------------------------
    clusters = {}
    for idx in range(n_edges):
        for a in best["orders"][idx]["ids"]:
            clusters.setdefault(int(a), []).append(idx)
    id2overlap = {}
    cluster_modifier = [dict() for _ in range(n_edges)]
    for a in sorted(clusters):
        get_cluster_anchors(cluster_modifier, id2overlap, steps, a, clusters[a])
    cover["multi_clique"] = sum(1 for k in id2overlap if k[1] > 0)

    # ---- per read: the anchors it carries, in read order, :721-752 -------------------------------------------------
    vertex_info = [[] for _ in range(n_edges + 1)]
    vertices = [None] * (n_edges + 1)
    match_modifiers = {}
    for idx in range(n_edges):
        for m in best["modifiers"][idx]:
            match_modifiers[m] = match_modifiers.get(m, 0) + 1
        ids = [int(x) for x in best["orders"][idx]["ids"]]
        if is_neg[best["orders"][idx]["base"]]:
            ids.reverse()
        va, vb = path[idx], path[idx + 1]
        for a in ids:
            match = ((a, cluster_modifier[idx][a]), match_modifiers.get(a, 0))
            ra, rb = vm[(va["id"], a)], vm[(vb["id"], a)]
            vertex_info[idx].append(((int(ra["n_lo"]), int(ra["n_hi"])), match))
            vertex_info[idx + 1].append(((int(rb["n_lo"]), int(rb["n_hi"])), match))
        vertices[idx], vertices[idx + 1] = va, vb

    # ---- anchor DAG, anchor sequences, flanks, :754-853 ------------------------------------------------------------
    registry = {}

    def reg(name):  # Registry::operator[], Registry.cpp:36-45
        return registry.setdefault(name, len(registry))

    adg = _Adg()
    reg2id, anchor_seq, nanopores, pre, post = {}, {}, {}, {}, {}
    for idx, v in enumerate(vertices):
        rid, pos, neg = v["id"], v["dir"] is True or v["dir"] == 1, is_neg[v["id"]]

        def cmp(lhs, rhs):  # :760-770
            if lhs[0] == rhs[0]:
                cover["nr_ties"] += lhs[1] != rhs[1]
                lo, ro = id2overlap[lhs[1][0]], id2overlap[rhs[1][0]]
                if not _vm(vm[(rid, lhs[1][0][0])])["direction"]:
                    return -1 if ro < lo else (1 if lo < ro else 0)
                return -1 if lo < ro else (1 if ro < lo else 0)
            return -1 if lhs[0] < rhs[0] else 1

        info = sorted(vertex_info[idx], key=functools.cmp_to_key(cmp))
        if neg:
            info.reverse()
        vertex_info[idx] = info
        if not info:
            continue

        def ensure(match):  # :787-793, :800-806
            name = tuple_tuple2id(match)
            known = name in registry
            r = reg(name)
            if not known or r not in adg.succ:
                adg.add_vertex(r)
                anchor_seq[r] = get_anchor_sequence(vm[(rid, match[0][0])], illu[match[0][0]], id2overlap[match[0]], pos)
                reg2id[r] = match[0]
            return r

        last_nr, last_match = info[0]
        for nr, match in info:
            r = ensure(match)
            if match == last_match:
                continue
            rl = ensure(last_match)
            flip = False
            if (last_nr[1] > nr[1] and last_nr[0] < nr[0]) or (last_nr[1] < nr[1] and last_nr[0] > nr[0]):  # :809-822
                cl = _corrected_range(_vm(vm[(rid, last_match[0][0])]), id2overlap[last_match[0]])
                cr = _corrected_range(_vm(vm[(rid, match[0][0])]), id2overlap[match[0]])
                flip = (pos and (cl[0] > cr[0] or (cl[0] == cr[0] and cl[1] > cr[1]))) or \
                       (neg and (cl[0] < cr[0] or (cl[0] == cr[0] and cl[1] < cr[1])))
            cover["flips"] += flip
            cover["dup_edges"] += ((r, rl) if flip else (rl, r)) in adg.edge_of
            e = adg.add_edge(r, rl) if flip else adg.add_edge(rl, r)
            nanopores.setdefault(e, []).append(v)
            last_match, last_nr = match, nr

        first, second = info[0][1], info[-1][1]
        pre.setdefault(registry[tuple_tuple2id(first)], []).append(get_sequence_left_of_anchor(
            vm[(rid, first[0][0])], nano[rid], illu[first[0][0]], v["len"], id2overlap[first[0]], pos))
        post.setdefault(registry[tuple_tuple2id(second)], []).append(get_sequence_right_of_anchor(
            vm[(rid, second[0][0])], nano[rid], illu[second[0][0]], v["len"], id2overlap[second[0]], pos))

    # ---- sequences between neighbouring anchors, :855-863 + alignAnchorRegion :581-611 -----------------------------
    distances, sequences = {}, {}
    for e, (u, w) in enumerate(adg.edges):
        a_l, a_r = reg2id[u][0], reg2id[w][0]
        dist, seqs = None, []
        for v in nanopores[e]:
            d, s = get_sequence_between_anchors(vm[(v["id"], a_l)], vm[(v["id"], a_r)], nano[v["id"]], illu[a_l],
                                                illu[a_r], id2overlap[reg2id[u]], id2overlap[reg2id[w]],
                                                v["dir"] is True or v["dir"] == 1)
            if s is not None:
                seqs.append(s)
            else:
                cover["no_seq"] += 1
            if dist is None:
                dist = d
        distances[e], sequences[e] = dist, seqs

    # ---- placement, :865-1010 ---------------------------------------------------------------------------------------
    order = adg.sort_topologically()
    if len(order) != len(adg.succ) or not order:
        raise AssemblyError("the anchor graph has a cycle (the reference throws std::out_of_range in a pool thread)")
    pos_of = {v: i for i, v in enumerate(order)}
    visited, tap = {}, {}
    glob = _visit_ordered(visited, tap, adg, reg2id, pos_of, order, distances, sequences, anchor_seq, id2overlap,
                          order[0])
    if len(adg.succ) == 1:  # :886-895
        a = adg.vertices()[0]
        ov = id2overlap[reg2id[a]]
        tap[a] = (0, ov[1] - ov[0])
        glob.seq, glob.lo, glob.hi = anchor_seq[a], 0, ov[1] - ov[0]

    additional = []
    for v in order[1:]:  # :897-925
        if v in visited:
            continue
        ltap = {}
        loc = _visit_ordered(visited, ltap, adg, reg2id, pos_of, order, distances, sequences, anchor_seq, id2overlap, v)
        if not ltap:
            ov = id2overlap[reg2id[v]]
            ltap[v] = (0, ov[1] - ov[0])
            loc.seq, loc.lo, loc.hi = anchor_seq[v], 0, ov[1] - ov[0]
        additional.append([loc, ltap, False])
    cover["extra_groups"] = len(additional)

    loop = True
    while loop:  # :927-1010
        loop = False
        progress = False
        for item in additional:
            if item[2]:
                continue
            loc, ltap = item[0].copy(), item[1]
            group_offset, found = 0, False
            for m in sorted(ltap):
                found = False
                for t in sorted(adg.succ[m]):
                    if t in tap:
                        e = adg.succ[m][t]
                        group_offset = tap[t][0] - distances[e] - ltap[m][1] - 1
                        if sequences[e]:
                            loc.update(sequences[e][0], ltap[m][1] + 1, ltap[m][1] + distances[e])
                        found = True
                        break
                if found:
                    break
                for t in sorted(adg.pred[m]):
                    if t in tap:
                        e = adg.pred[m][t]
                        group_offset = tap[t][1] + distances[e] + 1 - ltap[m][0] + 1
                        if sequences[e]:
                            loc.update(sequences[e][0], ltap[m][0] - distances[e], ltap[m][0] - 1)
                        found = True
                        break
                if found:
                    break
            if not found:
                loop = True
                continue
            item[2] = True
            progress = True
            for m in sorted(ltap):
                tap[m] = (ltap[m][0] + group_offset, ltap[m][1] + group_offset)
            glob.update(loc.seq, loc.lo + group_offset, loc.hi + group_offset)
        if loop and not progress:
            raise AssemblyError("a group of anchors never connects to the contig (the reference loops forever)")

    for v in adg.vertices():  # :1012-1032
        if v in pre:
            s = max(pre[v], key=len)  # std::max_element: the first of the longest
            glob.update(s, tap[v][0] - len(s), tap[v][0] - 1)
        if v in post:
            s = max(post[v], key=len)
            glob.update(s, tap[v][1] + 1, tap[v][1] + len(s))

    # ---- output records, :1034-1361 ---------------------------------------------------------------------------------
    left_most = -glob.lo
    tname = b"muchsalsa_%d" % asm_idx
    target_fa = b">" + tname + b"\n" + limit_length(glob.seq) + b"\n"
    tlen = len(glob.seq)
    queries, query_fa, paf = [], [], []

    def emit(kind, s, lb, rb):
        name = b"%s.%d.%d" % (kind, asm_idx, len(queries))
        queries.append((name, s, lb, rb))
        query_fa.append(b">" + name + b"\n" + limit_length(s) + b"\n")
        paf.append(b"%s\t%d\t0\t%d\t+\t%s\t%d\t%d\t%d\t%d\t%d\t255\n" % (name, len(s), len(s), tname, tlen, lb, rb,
                                                                          rb - lb + 1, rb - lb + 1))

    for e, (u, w) in enumerate(adg.edges):  # :1052-1109
        for s in sequences[e]:
            if not s:
                continue
            emit(b"Middle", s, tap[u][1] + 1 + left_most, tap[w][0] - 1 + left_most)
    for v in adg.vertices():  # :1111-1225
        for s in pre.get(v, []):
            if len(s) < TH_SEQUENCE_LENGTH:
                continue
            rb = tap[v][0] - 1 + left_most
            emit(b"Left", s, rb - len(s) + 1, rb)
        for s in post.get(v, []):
            if len(s) < TH_SEQUENCE_LENGTH:
                continue
            lb = tap[v][1] + 1 + left_most
            emit(b"Right", s, lb, lb + len(s) - 1)

    for idx, v in enumerate(vertices):  # :1227-1361
        id2anchor = {}
        for info in vertex_info[idx]:
            id2anchor[info[1][0][0]] = info[1]
        pos = v["dir"] is True or v["dir"] == 1
        for ce in contains.get(v["id"], []):
            cinfo = sorted(((int(r["n_lo"]), int(r["n_hi"])), int(a)) for a, r in ce["matches"].items()
                           if int(a) in id2anchor)
            if not cinfo:
                continue
            direction = bool(ce["dir"]) == pos
            if not direction:
                cinfo.reverse()
            ranges = []
            for nr, a in cinfo:
                tap_id = id2anchor[a]
                tap_dir = _vm(vm[(v["id"], a)])["direction"] == pos
                ov = id2overlap[tap_id[0]]
                illumina_ref = ov[1] if tap_dir else ov[0]
                total_ref = tap[registry[tuple_tuple2id(tap_id)]][1] + left_most
                cm = _vm(ce["matches"][a])
                cont_dir = cm["direction"] == direction
                ir = cm["illu"]
                if not cont_dir:
                    off = ir[0] - illumina_ref
                    ranges.append((total_ref - off - (ir[1] - ir[0]), total_ref - off))
                else:
                    off = ir[1] - illumina_ref
                    ranges.append((total_ref + off - (ir[1] - ir[0]), total_ref + off))
            to_write = []
            for k, (nr, a) in enumerate(cinfo):
                cm = _vm(ce["matches"][a])
                to_write.append((get_sequence(illu[a], cm["illu"][0], cm["illu"][1], cm["direction"] == direction),
                                 ranges[k][0], ranges[k][1], b"Illumina_Match"))
                if k == 0:
                    continue
                pre_n = cinfo[k - 1][0]
                to_write.append((get_sequence(nano[ce["nano"]], pre_n[1] + 1, cm["nano"][0] - 1, direction),
                                 ranges[k - 1][1] + 1, ranges[k][0] - 1, b"Nano_Middle"))
            for s, lb, rb, kind in to_write:
                if len(s) < TH_SEQUENCE_LENGTH:
                    continue
                emit(b"Contain_" + kind, s, lb, rb)
                cover["contain_records"] += 1

    return dict(target=glob.seq, target_fa=target_fa, query_fa=b"".join(query_fa), paf=b"".join(paf), queries=queries,
                id2overlap=id2overlap, tap=tap, left_most=left_most, borders=(glob.lo, glob.hi),
                n_anchors=len(adg.succ), n_anchor_edges=len(adg.edges), best=best, cover=cover)
