"""Inputs for assemblePath (ap.cpp:615-1362) built from the synthetic workload: sequences, overlap tables -> paths.

Test-side stand-in for the phases between the overlap path and assemblePath (graph clean-up, getDirectedGraph,
linearizeGraph -- SURVEY section 8 rows F1/F2, not built yet): a path is a greedy left-to-right chain of reads along
the synthetic genome whose consecutive reads share an overlap-graph edge with at least one EdgeOrder; the orders put on
the directed edge path[i] -> path[i+1] follow getDirectedGraph's flip rule (dg.cpp:75-84).
"""
import numpy as np

from muchsalsa_amd import synth

_COMP = bytes.maketrans(b"ACGT", b"TGCA")
ORD_START_V1, ORD_CONTAINED, ORD_DIR, ORD_PRIMARY = 1, 2, 4, 8


def revcomp(s):
    return s.translate(_COMP)[::-1]


class World:
    """Synthetic reads/anchors with sequences, accepted rows and a (read, anchor) -> row index."""

    def __init__(self, n_reads, read_len, n_anchors, seed, jitter=15, coverage=10):
        tab = synth.paf_table(n_reads, read_len, n_anchors, seed, coverage=coverage, jitter=jitter)
        self.rows, read_names, anchor_names = synth.accepted_rows(tab)
        G, r_start, r_fwd = synth.read_layout(n_reads, read_len, seed, coverage)
        a_start, a_len = synth.anchor_layout(n_reads, read_len, n_anchors, seed, coverage)
        self.genome = synth.genome_bases(G, seed).tobytes()
        self.read_orig = [int(n[1:]) for n in read_names]  # registry id -> generator index
        self.anchor_orig = [int(n[1:]) for n in anchor_names]
        self.read_start = [int(r_start[i]) for i in self.read_orig]
        self.read_fwd = [bool(r_fwd[i]) for i in self.read_orig]
        self.read_len = read_len
        self.nano, self.illu = {}, {}
        for rid, i in enumerate(self.read_orig):
            s = self.genome[int(r_start[i]): int(r_start[i]) + read_len]
            self.nano[rid] = s if r_fwd[i] else revcomp(s)
        for aid, j in enumerate(self.anchor_orig):
            self.illu[aid] = self.genome[int(a_start[j]): int(a_start[j]) + int(a_len[j])]
        # MatchMap::addVertexMatch keeps the lowest line per (read, anchor); the generator has no duplicates
        self.vm = {(int(r["read_id"]), int(r["anchor_id"])): r for r in self.rows}

    def attach(self, tables):
        """tables = result of the overlap path (oracle or product): edges, ems, orders, ids"""
        self.tables = tables
        self.edge_of = {}
        for i, e in enumerate(tables["edges"]):
            self.edge_of[(int(e["v1"]), int(e["v2"]))] = i
        self.adj = {}
        for (a, b) in self.edge_of:
            self.adj.setdefault(a, []).append(b)
            self.adj.setdefault(b, []).append(a)

    def edge(self, a, b):
        return self.edge_of.get((a, b), self.edge_of.get((b, a)))

    def step(self, a, b, dir_a):
        """orders + EdgeMatches of the directed edge a -> b as getDirectedGraph would fill it (dg.cpp:70-102)"""
        t = self.tables
        e = t["edges"][self.edge(a, b)]
        orders = []
        for o in t["orders"][int(e["order_off"]): int(e["order_off"]) + int(e["order_cnt"])]:
            flip = False
            if not (int(o["flags"]) & ORD_DIR) and int(o["base"]) == b:
                flip = not flip
            if not dir_a:
                flip = not flip
            start, end = int(o["start"]), int(o["end"])
            if flip:
                start, end = end, start
            if (start, end) != (a, b):
                continue
            ids = t["ids"][int(o["ids_off"]): int(o["ids_off"]) + int(o["ids_cnt"])]
            orders.append({"ids": [int(x) for x in ids], "score": int(o["score"]), "base": int(o["base"])})
        ems = t["ems"][int(e["em_off"]): int(e["em_off"]) + int(e["em_cnt"])]
        em = {int(m["anchor_id"]): (int(m["ov_lo"]), int(m["ov_hi"])) for m in ems}
        return {"orders": orders, "em": em}

    def chain(self, start_read, max_len=12, flip_all=False, dense=False):
        """greedy chain to the right of start_read -> (path, steps); dense: nearest neighbour instead of farthest"""
        def direction(r):
            return self.read_fwd[r] != flip_all

        path, steps, cur, used = [start_read], [], start_read, {start_read}
        while len(path) < max_len:
            best = None
            for nb in self.adj.get(cur, []):
                if nb in used or self.read_start[nb] <= self.read_start[cur]:
                    continue
                st = self.step(cur, nb, direction(cur)) if not flip_all else self.step(nb, cur, direction(nb))
                if not st["orders"]:
                    continue
                if best is None or (self.read_start[nb] > self.read_start[best[0]]) != dense:
                    best = (nb, st)
            if best is None:
                break
            path.append(best[0])
            steps.append(best[1])
            used.add(best[0])
            cur = best[0]
        if flip_all:  # walk the same chain right-to-left: every read has the opposite direction
            path.reverse()
            steps.reverse()
        return [{"id": r, "dir": direction(r), "len": self.read_len} for r in path], steps

    def contained_in(self, read):
        """ContainElements for `read`: neighbours whose only relation is a contained order based at them (test data)"""
        out = []
        t = self.tables
        for nb in self.adj.get(read, []):
            e = t["edges"][self.edge(read, nb)]
            for o in t["orders"][int(e["order_off"]): int(e["order_off"]) + int(e["order_cnt"])]:
                if int(o["flags"]) & ORD_CONTAINED:
                    ids = t["ids"][int(o["ids_off"]): int(o["ids_off"]) + int(o["ids_cnt"])]
                    out.append({"nano": nb, "dir": bool(int(o["flags"]) & ORD_DIR),
                                "matches": {int(a): self.vm[(nb, int(a))] for a in ids}})
                    break
        return out
