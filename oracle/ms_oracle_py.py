"""Second, independent CPU restatement of the reference overlap path, in pure Python (TEST INFRASTRUCTURE ONLY).

Written from the reference sources in the reference's own shape (dict-of-dict MatchMap, edge objects, vectors of
tuples), deliberately NOT from oracle/ms_oracle.c, so that the two restatements check each other: tests require the
C oracle and this module to agree bit for bit (ints exact, doubles via float.hex) on small seeded inputs.
PARITY UNPINNED by the reference's own tests (none covers this path) -- see oracle/ms_oracle.h.

Each function cites the reference lines it follows (paths relative to the reference tree).
Pure-Python loops: use on small inputs only (a few hundred reads).
"""


class VertexMatch:  # include/ms/matching/MatchMap.h:51-59
    __slots__ = ("nano", "illu", "r_ratio", "direction", "score", "is_primary", "line")

    def __init__(self, nano, illu, r_ratio, direction, score, is_primary, line):
        self.nano, self.illu, self.r_ratio = nano, illu, r_ratio
        self.direction, self.score, self.is_primary, self.line = direction, score, is_primary, line


class EdgeMatch:  # include/ms/matching/MatchMap.h:68-74
    __slots__ = ("overlap", "direction", "score", "is_primary", "line")

    def __init__(self, overlap, direction, score, is_primary, line):
        self.overlap, self.direction, self.score, self.is_primary, self.line = overlap, direction, score, is_primary, line


class Edge:  # include/ms/graph/Edge.h:212-218
    def __init__(self, v1, v2):
        self.vertices = (v1, v2)
        self.orders = []
        self.shadow = False


def parse_paf_text(text, min_matches=400, th_length=500, th_matches=500):
    """BlastFileAccessor::_buildIndex + BlastFileReader::read/parseLine (BlastFileReader.cpp:72-130)."""
    # line index: every '\n'-terminated line + a non-empty unterminated tail (IO.cpp:54-97)
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    reg_n, reg_i = {}, {}
    rows = []
    for line_idx in range(max(len(lines), 1) - 1):  # :76 -- the last line is never parsed
        tokens = lines[line_idx].split("\t")
        if tokens[-1] == "":
            tokens.pop()  # std::getline yields no token after the final delimiter (nor for an empty line)
        if len(tokens) < 10:
            raise ValueError("Invalid BLAST file.")
        illu = (int(tokens[2]), int(tokens[3]) - 1)
        matches = int(tokens[9])
        nano_len = int(tokens[6])
        add = matches >= min_matches and (illu[1] - illu[0] + 1) >= min_matches  # :106-107
        if not add:
            continue
        nid = reg_n.setdefault(tokens[5], len(reg_n))  # :110
        iid = reg_i.setdefault(tokens[0], len(reg_i))  # :111
        nano = (int(tokens[7]), int(tokens[8]) - 1)
        direction = tokens[4] == "+"
        prim = (illu[1] - illu[0] + 1) >= th_length and matches >= th_matches
        rows.append(dict(anchor_id=iid, read_id=nid, read_len=nano_len, i_lo=illu[0], i_hi=illu[1], n_lo=nano[0],
                         n_hi=nano[1], score=matches, line=line_idx, flags=(1 if direction else 0) | (2 if prim else 0)))
    return rows, list(reg_n), list(reg_i)


class MatchMap:
    """include/ms/matching/MatchMap.h:98-224 + the Graph vertices it observes."""

    def __init__(self):
        self.vertex_matches = {}  # nanoporeId -> illuminaId -> VertexMatch      (m_vertexMatches)
        self.scaffolds = {}       # illuminaId -> nanoporeId -> VertexMatch      (m_scaffolds)
        self.edge_matches = {}    # (v1, v2)   -> illuminaId -> EdgeMatch        (m_edgeMatches)
        self.vertices = {}        # nanoporeId -> (nanoporeLength, first line)   (Graph::m_vertices)
        self.edges = {}           # (v1, v2)   -> Edge

    def add_row(self, r):  # BlastFileReader.cpp:113-126
        nid, iid = r["read_id"], r["anchor_id"]
        self.vertices.setdefault(nid, (r["read_len"], r["line"]))  # Graph.cpp:148 emplace: first wins
        i_span = r["i_hi"] - r["i_lo"] + 1
        n_span = r["n_hi"] - r["n_lo"] + 1
        with_inf = float(i_span) / float(n_span) if n_span != 0 else float("inf") * (1 if i_span > 0 else -1)
        vm = VertexMatch((r["n_lo"], r["n_hi"]), (r["i_lo"], r["i_hi"]), with_inf, bool(r["flags"] & 1), r["score"],
                         bool(r["flags"] & 2), r["line"])
        ids = self.vertex_matches.setdefault(nid, {})  # MatchMap.cpp:52-81
        if iid in ids:
            insert = ids[iid].line > vm.line
        else:
            insert = True
        if insert:
            ids[iid] = vm
            self.scaffolds.setdefault(iid, {})[nid] = vm

    def calculate_edges(self, th_overlap=100):  # MatchMap.cpp:161-224
        for iid, scaffold in self.scaffolds.items():
            idx = sorted(((vm.line, nid) for nid, vm in scaffold.items()), key=lambda t: t[0])
            for o in range(1, len(idx)):
                outer = scaffold[idx[o][1]]
                for i in range(o):
                    inner = scaffold[idx[i][1]]
                    ov = (max(outer.illu[0], inner.illu[0]), min(outer.illu[1], inner.illu[1]))
                    if ov[0] <= ov[1] and ov[1] - ov[0] > th_overlap:
                        direction = outer.direction == inner.direction
                        prim = outer.is_primary and inner.is_primary
                        ol = float(outer.illu[1] - outer.illu[0] + 1)
                        il = float(inner.illu[1] - inner.illu[0] + 1)
                        cl = float(ov[1] - ov[0] + 1)
                        s = float(outer.score) * cl / ol + float(inner.score) * cl / il
                        vo, vi = idx[o][1], idx[i][1]
                        if self.vertices[vo][1] < self.vertices[vi][1]:  # :204-213
                            key = (vo, vi)
                        else:
                            key = (vi, vo)
                        self.edges.setdefault(key, Edge(*key))
                        ems = self.edge_matches.setdefault(key, {})
                        em = EdgeMatch(ov, direction, s, prim, outer.line)
                        if iid in ems:  # MatchMap.cpp:109-134
                            if ems[iid].line > em.line:
                                ems[iid] = em
                        else:
                            ems[iid] = em


def _div(a, b):
    """IEEE double division including x/inf and x/0 (Python raises on /0)."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0:
            return float("nan")
        return float("inf") if (a > 0) == (str(b)[0] != "-") else float("-inf")


def _std_max(a, b):
    return b if a < b else a


def _std_min(a, b):
    return b if b < a else a


def check_compatibility(mm, edge, id1, id2, wiggle, ratio_pct=15):  # mpp.cpp:38-142
    def nano_check(vertex):
        em1, em2 = mm.edge_matches[edge.vertices][id1], mm.edge_matches[edge.vertices][id2]
        vm1, vm2 = mm.vertex_matches[vertex][id1], mm.vertex_matches[vertex][id2]
        ncl1 = _div(float(em1.overlap[0] - vm1.illu[0]), vm1.r_ratio)
        ncr1 = _div(float(vm1.illu[1] - em1.overlap[1]), vm1.r_ratio)
        if not vm1.direction:
            ncl1, ncr1 = ncr1, ncl1
        ncl2 = _div(float(em2.overlap[0] - vm2.illu[0]), vm2.r_ratio)
        ncr2 = _div(float(vm2.illu[1] - em2.overlap[1]), vm2.r_ratio)
        if not vm2.direction:
            ncl2, ncr2 = ncr2, ncl2
        c1 = (float(vm1.nano[0]) + ncl1, float(vm1.nano[1]) - ncr1)
        c2 = (float(vm2.nano[0]) + ncl2, float(vm2.nano[1]) - ncr2)
        orientation, diff = 0, 0.0
        if c1[0] <= c2[1] and c2[0] <= c1[1]:
            if c1[0] < c2[0] and c1[1] < c2[1]:
                orientation, diff = 2, c1[1] - c2[0] + 1
            if c1[0] > c2[0] and c1[1] > c2[1]:
                orientation, diff = -2, c2[1] - c1[0] + 1
        elif c1[0] < c2[0]:
            orientation, diff = 1, c2[0] - c1[1] + 1
        else:
            orientation, diff = -1, c1[0] - c2[1] + 1
        uco = 0
        if vm1.nano[0] <= vm2.nano[1] and vm2.nano[0] <= vm1.nano[1]:
            if vm1.nano[0] < vm2.nano[0] and vm1.nano[1] < vm2.nano[1]:
                uco = 2
            if vm1.nano[0] > vm2.nano[0] and vm1.nano[1] > vm2.nano[1]:
                uco = -2
            if (orientation < 0 and uco >= 0) or (orientation > 0 and uco <= 0):
                return True, orientation, diff
        return False, orientation, diff

    a1, o1, d1 = nano_check(edge.vertices[0])
    a2, o2, d2 = nano_check(edge.vertices[1])
    if a1 or a2:
        return False
    if not mm.edge_matches[edge.vertices][id1].direction:
        o2 = -o2
    if o1 == o2 and o1 != 0:
        diff = _std_max(d1, d2) - _std_min(d1, d2)
        return diff <= float(wiggle) or _div(diff * 100, _std_max(d1, d2)) <= ratio_pct
    if (o1 < 0 and o2 < 0) or (o1 > 0 and o2 > 0):
        return d1 + d2 <= float(wiggle)
    return False


def get_max_pairwise_paths(mm, edge, illumina_ids, direction, wiggle, counters=None, alt_frac=0.75):  # mpp.cpp:145-305
    result = []
    if not illumina_ids:
        return result
    v1, v2 = edge.vertices
    ems = mm.edge_matches[edge.vertices]
    v_start = sorted((mm.vertex_matches[v1][i].nano, i) for i in illumina_ids)
    population = [([], ems[t[1]].score) for t in v_start]
    limit = max(len(v_start), 1) - 1
    for k in range(limit):
        for l in range(k + 1, limit + 1):
            if counters is not None:
                counters["compat"] = counters.get("compat", 0) + 1
            ok = check_compatibility(mm, edge, v_start[k][1], v_start[l][1], wiggle)
            score = population[k][1] + ems[v_start[l][1]].score
            ok = ok and score > population[l][1]
            if ok:
                population[l] = (population[k][0] + [k], score)
    max_val, max_idx = 0.0, 0
    for i in range(len(population)):
        population[i][0].append(i)
        if population[i][1] > max_val:
            max_idx, max_val = i, population[i][1]
    v_max = population[max_idx][0]
    has_primary = any(ems[v_start[i][1]].is_primary for i in v_max) or len(v_max) > 2
    result.append(([v_start[i][1] for i in v_max], int(max_val), has_primary))
    thr = max_val * alt_frac
    for path, score in population:
        if score > thr:
            ids = [v_start[i][1] for i in path]
            if all(i not in r[0] for r in result for i in ids):
                result.append((ids, int(score), any(ems[i].is_primary for i in ids)))
    if len(result) == 1 and result[0][2]:
        s_list = sorted((vm.nano, i) for i, vm in mm.vertex_matches[v1].items())
        e_list = sorted((vm.nano, i) for i, vm in mm.vertex_matches[v2].items())
        if not direction:
            e_list.reverse()
        p_ids = result[0][0]
        if (s_list[0][1] != p_ids[0] and e_list[0][1] != p_ids[0]) or \
                (s_list[-1][1] != p_ids[-1] and e_list[-1][1] != p_ids[-1]):
            result = [(p_ids, result[0][1], False)]
        else:
            def find_from(lst, start, ident):  # std::find_if; start beyond end() returns end() (libstdc++)
                for q in range(start, len(lst)):
                    if lst[q][1] == ident:
                        return q
                return len(lst)
            i = j = 0
            is_shadow = False
            for ident in p_ids:
                if is_shadow:
                    break
                rs = find_from(s_list, i, ident)
                inter = rs > i
                i += (rs - i) + 1
                re_ = find_from(e_list, j, ident)
                inter = inter and re_ > j
                j += (re_ - j) + 1
                is_shadow = inter
            if is_shadow:
                result = [(p_ids, result[0][1], False)]
    return result


def get_overhangs(mm, vertex, edge, iid):  # ol.cpp:31-50
    vm = mm.vertex_matches[vertex][iid]
    em = mm.edge_matches[edge.vertices][iid]
    ncl = _div(float(em.overlap[0] - vm.illu[0]), vm.r_ratio)
    ncr = _div(float(vm.illu[1] - em.overlap[1]), vm.r_ratio)
    if not vm.direction:
        ncl, ncr = ncr, ncl
    left = float(vm.nano[0]) + ncl
    length = int(mm.vertices[vertex][0])
    right = float(length - vm.nano[1]) + ncr
    return left, right


def get_overlap(mm, ids, edge, direction, score, is_primary):  # ol.cpp:53-101
    v1, v2 = edge.vertices
    f1 = get_overhangs(mm, v1, edge, ids[0])
    l1 = get_overhangs(mm, v1, edge, ids[-1])
    f2 = get_overhangs(mm, v2, edge, ids[0])
    l2 = get_overhangs(mm, v2, edge, ids[-1])
    L1, R1, L2, R2 = f1[0], l1[1], f2[0], l2[1]
    if not direction:
        L2, R2 = f2[1], l2[0]
    if L1 <= L2 and R1 <= R2:
        return dict(start=v1, end=v2, left=L2 - L1, right=R2 - R1, contained=True, base=v1, score=score, ids=ids,
                    direction=direction, primary=is_primary)
    if L1 >= L2 and R1 >= R2:
        return dict(start=v2, end=v1, left=L1 - L2, right=R1 - R2, contained=True, base=v1, score=score, ids=ids,
                    direction=direction, primary=is_primary)
    if L1 > L2 and R1 < R2:
        return dict(start=v1, end=v2, left=L1 - L2, right=R2 - R1, contained=False, base=v1, score=score, ids=ids,
                    direction=direction, primary=is_primary)
    if L1 < L2 and R1 > R2:
        return dict(start=v2, end=v1, left=L2 - L1, right=R1 - R2, contained=False, base=v1, score=score, ids=ids,
                    direction=direction, primary=is_primary)
    return None


def chaining_and_overlaps(mm, edge, wiggle=300, counters=None):  # src/main.cpp:328-414
    ems = mm.edge_matches.get(edge.vertices)
    if not ems:
        return
    plus = [i for i, em in ems.items() if em.direction]
    minus = [i for i, em in ems.items() if not em.direction]
    minus_paths = get_max_pairwise_paths(mm, edge, minus, False, wiggle, counters)
    plus_paths = get_max_pairwise_paths(mm, edge, plus, True, wiggle, counters)
    has_primary = any(p[2] for p in plus_paths) or any(p[2] for p in minus_paths)
    if has_primary:
        plus_paths = [p for p in plus_paths if p[2]]
        minus_paths = [p for p in minus_paths if p[2]]
    has_multi = any(len(p[0]) > 1 for p in plus_paths) or any(len(p[0]) > 1 for p in minus_paths)
    if has_multi:
        plus_paths = [p for p in plus_paths if len(p[0]) > 1]
        minus_paths = [p for p in minus_paths if len(p[0]) > 1]
    if len(plus_paths) + len(minus_paths) > 1:
        edge.shadow = True
    else:
        path = minus_paths[0] if minus_paths else plus_paths[0]
        edge.shadow = not path[2]
    for p in minus_paths:
        o = get_overlap(mm, p[0], edge, False, p[1], p[2])
        if o is not None:
            edge.orders.append(o)
    for p in plus_paths:
        o = get_overlap(mm, p[0], edge, True, p[1], p[2])
        if o is not None:
            edge.orders.append(o)


def overlap(rows, th_overlap=100, wiggle=300):
    """rows: iterable of dicts/structured rows with the ms_row fields.  Returns canonical python tables."""
    mm = MatchMap()
    for r in sorted(({k: int(r[k]) for k in ("anchor_id", "read_id", "read_len", "i_lo", "i_hi", "n_lo", "n_hi", "score",
                                             "line", "flags")} for r in rows), key=lambda r: r["line"]):
        mm.add_row(r)
    mm.calculate_edges(th_overlap)
    counters = {}
    for e in mm.edges.values():
        chaining_and_overlaps(mm, e, wiggle, counters)
    edges = []
    for key in sorted(mm.edges):
        e = mm.edges[key]
        v1 = key[0]
        ems = sorted(mm.edge_matches[key].items(), key=lambda kv: (mm.vertex_matches[v1][kv[0]].nano, kv[0]))
        edges.append(dict(
            v1=key[0], v2=key[1], shadow=e.shadow,
            ems=[dict(anchor_id=i, ov_lo=m.overlap[0], ov_hi=m.overlap[1], score=m.score,
                      flags=(1 if m.direction else 0) | (2 if m.is_primary else 0), line=m.line) for i, m in ems],
            orders=e.orders))
    return edges, counters


# ----------------------------------------------------------------------------------------------------------------------
# sequence helpers and the segment builders of assemblePath, second restatement (strings, like the reference)
# ----------------------------------------------------------------------------------------------------------------------

def str_slice(s, i, j):  # SequenceUtils.cpp:27-38
    size = len(s)
    i2 = i if i >= 0 else size + i
    j2 = j if j >= 0 else size + j
    start = max(0, i2)
    end = max(min(size, max(0, j2)), i2 % (1 << 64))
    if start > size:
        raise IndexError("std::out_of_range")
    return s[start:start + (end - start + 1)]


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def reverse_complement(s):  # SequenceUtils.cpp:41-61
    return s.translate(_COMP)[::-1]


def get_sequence(seq, left, right, direction):  # getIlluminaSequence / getNanoporeSequence, SequenceUtils.cpp:63-85
    s = str_slice(seq, left, right + 1)
    return s if direction else reverse_complement(s)


def _vm(m):
    i_span = int(m["i_hi"]) - int(m["i_lo"]) + 1
    n_span = int(m["n_hi"]) - int(m["n_lo"]) + 1
    return dict(illu=(int(m["i_lo"]), int(m["i_hi"])), nano=(int(m["n_lo"]), int(m["n_hi"])),
                r_ratio=_div(float(i_span), float(n_span)), direction=bool(int(m["flags"]) & 1))


def get_anchor_sequence(m, illu, ov, direction):  # ap.cpp:424-433
    v = _vm(m)
    return get_sequence(illu, ov[0], ov[1], v["direction"] == bool(direction))


def get_sequence_left_of_anchor(m, nano, illu, nanopore_length, ov, direction):  # ap.cpp:352-386
    v = _vm(m)
    if not direction:
        if not v["direction"]:
            s = get_sequence(illu, v["illu"][0], ov[0], False)
        else:
            s = get_sequence(illu, ov[1], v["illu"][1], True)
        s += get_sequence(nano, v["nano"][1], int(nanopore_length) - 1, True)
        return reverse_complement(s)
    s = get_sequence(nano, 0, v["nano"][0], True)
    if not v["direction"]:
        s += get_sequence(illu, ov[1], v["illu"][1], False)
    else:
        s += get_sequence(illu, v["illu"][0], ov[0], True)
    return s


def get_sequence_right_of_anchor(m, nano, illu, nanopore_length, ov, direction):  # ap.cpp:388-422
    v = _vm(m)
    if not direction:
        s = get_sequence(nano, 0, v["nano"][0], True)
        if not v["direction"]:
            s += get_sequence(illu, ov[1], v["illu"][1], False)
        else:
            s += get_sequence(illu, v["illu"][0], ov[0], True)
        return reverse_complement(s)
    if not v["direction"]:
        s = get_sequence(illu, v["illu"][0], ov[0], False)
    else:
        s = get_sequence(illu, ov[1], v["illu"][1], True)
    return s + get_sequence(nano, v["nano"][1], int(nanopore_length) - 1, True)


def _corrected_range(v, ov):  # getCorrectedNanoporeRange, ap.cpp:191-203
    left = _div(float(ov[0] - v["illu"][0]), v["r_ratio"])
    right = _div(float(v["illu"][1] - ov[1]), v["r_ratio"])
    if not v["direction"]:
        left, right = right, left
    return v["nano"][0] + left, v["nano"][1] - right


def get_sequence_between_anchors(ml, mr, nano, illu_l, illu_r, ov_l, ov_r, direction):  # ap.cpp:435-579
    import math
    L, R = _vm(ml), _vm(mr)
    corr_l = corr_r = 0
    if not direction:
        err = float(R["nano"][1] - L["nano"][0])
        if err > 0:
            cl, cr = _corrected_range(L, ov_l), _corrected_range(R, ov_r)
            if cl[0] < cr[1]:
                return int(math.floor(cl[0] - cr[1])), None
            if not L["direction"]:
                avail, corr_l = _div(float(L["illu"][1] - ov_l[1]), L["r_ratio"]), L["illu"][1] - ov_l[1]
            else:
                avail, corr_l = _div(float(ov_l[0] - L["illu"][0]), L["r_ratio"]), ov_l[0] - L["illu"][0]
            if avail > err:
                corr_l, err = int(math.floor(err * L["r_ratio"])), 0.0
            else:
                err -= avail
            if not R["direction"]:
                avail, corr_r = _div(float(ov_r[0] - R["illu"][0]), R["r_ratio"]), ov_r[0] - R["illu"][0]
            else:
                avail, corr_r = _div(float(R["illu"][1] - ov_r[1]), R["r_ratio"]), R["illu"][1] - ov_r[1]
            if avail > err:
                corr_r = int(math.floor(err * R["r_ratio"]))
        if not R["direction"]:
            s = get_sequence(illu_r, R["illu"][0] + corr_r, ov_r[0], False)
        else:
            s = get_sequence(illu_r, ov_r[1], R["illu"][1] - corr_r, True)
        s += get_sequence(nano, R["nano"][1], L["nano"][0], True)
        if not L["direction"]:
            s += get_sequence(illu_l, ov_l[1], L["illu"][1] - corr_l, False)
        else:
            s += get_sequence(illu_l, L["illu"][0] + corr_l, ov_l[0], True)
        return len(s), reverse_complement(s)
    err = float(L["nano"][1] - R["nano"][0])
    if err > 0:
        cl, cr = _corrected_range(L, ov_l), _corrected_range(R, ov_r)
        if cl[1] > cr[0]:
            return int(math.floor(cr[0] - cl[1])), None
        if not L["direction"]:
            avail, corr_l = _div(float(ov_l[0] - L["illu"][0]), L["r_ratio"]), ov_l[0] - L["illu"][0]
        else:
            avail, corr_l = _div(float(L["illu"][1] - ov_l[1]), L["r_ratio"]), L["illu"][1] - ov_l[1]
        if avail > err:
            corr_l, err = int(math.floor(err * L["r_ratio"])), 0.0
        else:
            err -= avail
        if not R["direction"]:
            avail, corr_r = _div(float(R["illu"][1] - ov_r[1]), R["r_ratio"]), R["illu"][1] - ov_r[1]
        else:
            avail, corr_r = _div(float(ov_r[0] - R["illu"][0]), R["r_ratio"]), ov_r[0] - R["illu"][0]
        if avail > err:
            corr_r = int(math.floor(err * R["r_ratio"]))
    if not L["direction"]:
        s = get_sequence(illu_l, L["illu"][0] + corr_l, ov_l[0], False)
    else:
        s = get_sequence(illu_l, ov_l[1], L["illu"][1] - corr_l, True)
    s += get_sequence(nano, L["nano"][1], R["nano"][0], True)
    if not R["direction"]:
        s += get_sequence(illu_r, ov_r[1], R["illu"][1] - corr_r, False)
    else:
        s += get_sequence(illu_r, R["illu"][0] + corr_r, ov_r[0], True)
    return len(s), s
