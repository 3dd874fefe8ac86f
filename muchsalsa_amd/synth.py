"""Synthetic unitig-anchor -> long-read PAF generator (SURVEY.md section 8(d) workload shape).

Portable and deterministic: every random draw comes from a counter-based splitmix64 stream, so the same
(seed, shape) gives the same table with any numpy version and in any other language.

Genome of length G = reads * read_len / coverage; reads start ~U[0, G-L], strand ~Bernoulli(1/2), length exactly L,
named r<i>; anchors are independent unitigs, length ~U{500..1500}, start ~U[0, G-len], named u<i>; one PAF row per
(anchor, read) whose genomic intersection is >= 420 bp: query coords = intersection in the anchor frame, target
coords = intersection on the read's own strand, each end jittered by U{-15..15} and clamped to [0, L],
nmatch = floor(span * U(0.86, 0.97)); rows grouped by anchor then read; one trailing sentinel line (the reference
never parses the last line of the file, BlastFileReader.cpp:76).

`paf_table` returns PAF-level columns; `accepted_rows` applies the reference's A1 rules (filter + Registry ids,
BlastFileReader.cpp:101-126) and returns the msgpu_row table that msgpu_parse_paf would produce from the text.
"""
import numpy as np

ROW_DTYPE = np.dtype([("anchor_id", "<u4"), ("read_id", "<u4"), ("read_len", "<i4"), ("i_lo", "<i4"),
                      ("i_hi", "<i4"), ("n_lo", "<i4"), ("n_hi", "<i4"), ("score", "<u4"), ("line", "<u4"),
                      ("flags", "<u4")])

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, stream, n):
    """n outputs of splitmix64 started at state seed ^ (stream * 0xD1342543DE82EF95), as uint64."""
    with np.errstate(over="ignore"):
        s0 = np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
        z = s0 + (np.arange(1, n + 1, dtype=np.uint64) * _GAMMA)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _randint(seed, stream, n, lo, hi):
    """integers in [lo, hi] (element-wise bounds allowed)"""
    u = splitmix64(seed, stream, n)
    span = (np.asarray(hi, dtype=np.int64) - np.asarray(lo, dtype=np.int64) + 1).astype(np.uint64)
    return np.asarray(lo, dtype=np.int64) + (u % span).astype(np.int64)


def _uniform(seed, stream, n, lo, hi):
    u = (splitmix64(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def read_lengths(n_reads, read_len, seed, read_len_min=None):
    """per-read lengths: exactly read_len (the BASELINE shape), or ~U{read_len_min..read_len} (reads of mixed length, so
    that short reads lie inside long ones: contained EdgeOrders, the input of findContractionEdges)"""
    if read_len_min is None or read_len_min >= read_len:
        return np.full(n_reads, read_len, dtype=np.int64)
    return _randint(seed, 9, n_reads, read_len_min, read_len)


def read_layout(n_reads, read_len, seed, coverage=10, read_len_min=None):
    """(genome length, read starts, read strands) -- the same draws paf_table() makes."""
    G = max(int(n_reads) * int(read_len) // int(coverage), read_len + 1500)
    r_start = _randint(seed, 3, n_reads, 0, G - read_lengths(n_reads, read_len, seed, read_len_min))
    r_fwd = (splitmix64(seed, 4, n_reads) & np.uint64(1)).astype(bool)
    return G, r_start, r_fwd


def anchor_layout(n_reads, read_len, n_anchors, seed, coverage=10, tiled=False):
    """(anchor starts, anchor lengths) on the genome -- the same draws paf_table() makes.  tiled: the anchors are
    consecutive, NON-overlapping stretches of length ~U{500..1500} separated by gaps ~U{0..300} that cover the genome end to
    end -- what the unitigs of one genome are (n_anchors is ignored; about G / 1150 of them)."""
    G = max(int(n_reads) * int(read_len) // int(coverage), read_len + 1500)
    if tiled:
        n_max = G // 500 + 2
        a_len = _randint(seed, 1, n_max, 500, 1500)
        gap = _randint(seed, 10, n_max, 0, 300)
        a_start = np.cumsum(a_len + gap) - a_len - gap + _randint(seed, 11, 1, 0, 199)[0]
        keep = a_start + a_len <= G
        return a_start[keep], a_len[keep]
    a_len = _randint(seed, 1, n_anchors, 500, 1500)
    a_start = _randint(seed, 2, n_anchors, 0, G - a_len)
    return a_start, a_len


def genome_bases(G, seed):
    """Uniform ACGT genome of length G as a uint8 array (32 bases per splitmix64 draw)."""
    u = splitmix64(seed, 8, (G + 31) // 32)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((u[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)[:G]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes]


def paf_table(n_reads, read_len, n_anchors, seed, coverage=10, min_intersection=420, jitter=15, tiled=False,
              read_len_min=None):
    """PAF-level columns of the synthetic alignment set (before the reference's filter).  tiled / read_len_min: see
    anchor_layout / read_lengths (defaults = the BASELINE shape: independent random anchors, reads of exactly read_len)."""
    G, r_start, r_fwd = read_layout(n_reads, read_len, seed, coverage, read_len_min)
    a_start, a_len = anchor_layout(n_reads, read_len, n_anchors, seed, coverage, tiled)
    n_anchors = len(a_start)
    read_len = read_lengths(n_reads, read_len, seed, read_len_min)  # per read from here on

    order = np.argsort(a_start, kind="stable")
    starts = a_start[order]
    lo = np.searchsorted(starts, r_start - 1500, side="left")
    hi = np.searchsorted(starts, r_start + read_len, side="right")
    cnt = hi - lo
    tot = int(cnt.sum())
    rid = np.repeat(np.arange(n_reads, dtype=np.int64), cnt)
    first = np.repeat(np.cumsum(cnt) - cnt, cnt)
    k = np.arange(tot, dtype=np.int64) - first + np.repeat(lo, cnt)
    aid = order[k]
    g_lo = np.maximum(a_start[aid], r_start[rid])
    g_hi = np.minimum(a_start[aid] + a_len[aid], r_start[rid] + read_len[rid])
    keep = (g_hi - g_lo) >= min_intersection
    aid, rid, g_lo, g_hi = aid[keep], rid[keep], g_lo[keep], g_hi[keep]
    # rows grouped by anchor id, then read id
    o = np.lexsort((rid, aid))
    aid, rid, g_lo, g_hi = aid[o], rid[o], g_lo[o], g_hi[o]
    n = len(aid)
    q_lo = g_lo - a_start[aid]
    q_hi = g_hi - a_start[aid]
    fwd = r_fwd[rid]
    t_lo = np.where(fwd, g_lo - r_start[rid], r_start[rid] + read_len[rid] - g_hi)
    t_hi = np.where(fwd, g_hi - r_start[rid], r_start[rid] + read_len[rid] - g_lo)
    t_lo = np.maximum(0, t_lo + _randint(seed, 5, n, -jitter, jitter))
    t_hi = np.minimum(read_len[rid], t_hi + _randint(seed, 6, n, -jitter, jitter))
    nmatch = np.floor((q_hi - q_lo) * _uniform(seed, 7, n, 0.86, 0.97)).astype(np.int64)
    return {"qname_id": aid, "qlen": a_len[aid], "qstart": q_lo, "qend": q_hi, "strand": fwd, "tname_id": rid,
            "tlen": read_len[rid], "tstart": t_lo, "tend": t_hi, "nmatch": nmatch,
            "genome": G}


def accepted_rows(tab, min_matches=400, th_length=500, th_matches=500):
    """The reference's A1 rules applied to paf_table() columns -> msgpu_row table (line order)."""
    span = tab["qend"] - tab["qstart"]
    ok = (tab["nmatch"] >= min_matches) & (span >= min_matches)
    line = np.nonzero(ok)[0]

    def registry(names):  # dense ids in first-seen order (Registry.cpp:36-45)
        uniq, first_idx, inv = np.unique(names, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[np.argsort(first_idx, kind="stable")] = np.arange(len(uniq))
        return rank[inv], uniq[np.argsort(first_idx, kind="stable")]

    read_id, read_names = registry(tab["tname_id"][ok])
    anchor_id, anchor_names = registry(tab["qname_id"][ok])
    rows = np.zeros(len(line), dtype=ROW_DTYPE)
    rows["anchor_id"] = anchor_id
    rows["read_id"] = read_id
    rows["read_len"] = tab["tlen"][ok]
    rows["i_lo"] = tab["qstart"][ok]
    rows["i_hi"] = tab["qend"][ok] - 1
    rows["n_lo"] = tab["tstart"][ok]
    rows["n_hi"] = tab["tend"][ok] - 1
    rows["score"] = tab["nmatch"][ok]
    rows["line"] = line
    prim = (span[ok] >= th_length) & (tab["nmatch"][ok] >= th_matches)
    rows["flags"] = tab["strand"][ok].astype(np.uint32) | (prim.astype(np.uint32) << 1)
    return rows, ["r%d" % i for i in read_names], ["u%d" % i for i in anchor_names]


def paf_lines(tab):
    """PAF text lines (12 columns) + the trailing sentinel line the reference never parses."""
    out = []
    for i in range(len(tab["qname_id"])):
        out.append("u%d\t%d\t%d\t%d\t%s\tr%d\t%d\t%d\t%d\t%d\t%d\t60" % (
            tab["qname_id"][i], tab["qlen"][i], tab["qstart"][i], tab["qend"][i], "+" if tab["strand"][i] else "-",
            tab["tname_id"][i], tab["tlen"][i], tab["tstart"][i], tab["tend"][i], tab["nmatch"][i],
            tab["qend"][i] - tab["qstart"][i]))
    out.append("u0\t1\t0\t1\t+\tr0\t1\t0\t1\t0\t1\t0")
    return out


def synth_rows(n_reads, read_len, n_anchors, seed, coverage=10, **shape):
    """Convenience: accepted msgpu_row table of a synthetic workload."""
    rows, _, _ = accepted_rows(paf_table(n_reads, read_len, n_anchors, seed, coverage, **shape))
    return rows


ORD_DIR = 4  # msgpu_order.flags: EdgeOrder::direction


def directed_step(tables, edge_idx, a, b, dir_a):
    """EdgeOrders + EdgeMatches of the directed edge a -> b as getDirectedGraph fills it (dg.cpp:70-102): an order of
    the undirected edge goes onto (start, end), swapped when exactly one of {!order.direction and base == b, !dir_a}."""
    e = tables["edges"][edge_idx]
    orders = []
    for o in tables["orders"][int(e["order_off"]): int(e["order_off"]) + int(e["order_cnt"])]:
        flip = (not (int(o["flags"]) & ORD_DIR) and int(o["base"]) == b) != (not dir_a)
        start, end = (int(o["end"]), int(o["start"])) if flip else (int(o["start"]), int(o["end"]))
        if (start, end) != (a, b):
            continue
        ids = tables["ids"][int(o["ids_off"]): int(o["ids_off"]) + int(o["ids_cnt"])]
        orders.append({"ids": [int(x) for x in ids], "score": int(o["score"]), "base": int(o["base"])})
    ems = tables["ems"][int(e["em_off"]): int(e["em_off"]) + int(e["em_cnt"])]
    return {"orders": orders, "em": {int(m["anchor_id"]): (int(m["ov_lo"]), int(m["ov_hi"])) for m in ems}}


def chain_paths(tables, read_start, read_fwd, read_len, window_end, max_reads=12, max_paths=1 << 30):
    """Disjoint greedy left-to-right chains of reads starting before `window_end` on the synthetic genome.

    Stands in for the phases between the overlap path and assemblePath (graph clean-up, getDirectedGraph,
    linearizeGraph): consecutive reads share an overlap-graph edge with at least one
    EdgeOrder pointing along the chain; read direction = its strand.  read_start / read_fwd are indexed by read id.
    -> [(path, steps)] in the form muchsalsa_amd.assembly.Assembly.add_path takes."""
    edges = tables["edges"]
    inside = (read_start[edges["v1"]] < window_end) & (read_start[edges["v2"]] < window_end)
    adj = {}
    for i in np.nonzero(inside)[0]:
        a, b = int(edges["v1"][i]), int(edges["v2"][i])
        adj.setdefault(a, []).append((b, int(i)))
        adj.setdefault(b, []).append((a, int(i)))
    reads = [int(r) for r in np.argsort(read_start, kind="stable") if read_start[r] < window_end]
    used, out, frontier = set(), [], -1
    for s in reads:
        if s in used or read_start[s] <= frontier or len(out) >= max_paths:
            continue
        path, steps, cur = [s], [], s
        used.add(s)
        while len(path) < max_reads:
            best = None
            for nb, ei in adj.get(cur, []):
                if nb in used or read_start[nb] <= read_start[cur]:
                    continue
                if best is not None and read_start[nb] <= read_start[best[0]]:
                    continue
                st = directed_step(tables, ei, cur, nb, bool(read_fwd[cur]))
                if st["orders"]:
                    best = (nb, st)
            if best is None:
                break
            path.append(best[0])
            steps.append(best[1])
            used.add(best[0])
            cur = best[0]
        if len(path) >= 2:
            out.append(([{"id": r, "dir": bool(read_fwd[r]), "len": int(read_len)} for r in path], steps))
            frontier = int(read_start[path[-1]])
    return out


# the configurations BASELINE.json names
CONFIGS = {
    "cfg2": dict(n_reads=10_000, read_len=5_000, n_anchors=50_000, seed=42),
    "cfg3": dict(n_reads=100_000, read_len=10_000, n_anchors=500_000, seed=43),
}
# the same sizes on the shape the graph stage and assemblePath exist for: unitigs that TILE the genome (no two anchors
# overlap) and reads of mixed length (short ones contained in long ones).  Not BASELINE configurations.
TILED = {
    "cfg2": dict(n_reads=10_000, read_len=5_000, n_anchors=0, seed=42, tiled=True, read_len_min=1_250),
    "cfg3": dict(n_reads=100_000, read_len=10_000, n_anchors=0, seed=43, tiled=True, read_len_min=2_500),
}
