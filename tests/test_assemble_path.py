"""assemblePath (ap.cpp:615-1362): the product's host layout (copy pieces, layout-only context, no GPU) against the
Python restatement in oracle/ms_assemble_py.py, on chains of the synthetic workload and on fuzzed variants of them that
reach every branch of the function (several EdgeOrders per path edge, kinks, split anchor cliques, flipped anchor
pairs, equal nanopore ranges, detached anchor groups, reads without a sequence between anchors, contained reads)."""
import copy

import numpy as np
import pytest

from asmcases import World, revcomp
from segcases import apply_pieces

from muchsalsa_amd import _lib
from muchsalsa_amd.assembly import Assembly
from muchsalsa_amd.overlap import MsgpuError
from muchsalsa_amd.sequences import ILLUMINA, NANOPORE, SeqFile, SeqStore


@pytest.fixture(scope="module")
def world(oracle, tmp_path_factory):
    w = World(300, 5000, 1500, 7, jitter=15)
    w.attach(oracle.overlap(w.rows))
    d = tmp_path_factory.mktemp("asm")
    for name, seqs in (("n.fa", w.nano), ("i.fa", w.illu)):
        with open(d / name, "wb") as f:
            for i in range(len(seqs)):
                f.write(b">s%d\n" % i + seqs[i] + b"\n")
    w.files = (SeqFile(str(d / "n.fa")), SeqFile(str(d / "i.fa")))
    w.store = SeqStore(device=-1)  # layout-only: offsets and lengths, nothing can be gathered
    w.store.upload(NANOPORE, w.files[0])
    w.store.upload(ILLUMINA, w.files[1])
    # what a device store would hold: the loader's buffer as it is (a file parsed in several stretches leaves unused
    # bytes between them, so this is not the concatenation of the records)
    w.flat = (w.files[0].buffer(), w.files[1].buffer())
    assert all(w.files[0].sequence(i) == w.nano[i] for i in range(len(w.nano)))
    return w


def fuzz_case(w, rng, trial):
    """One (path, steps, rows, vm, contains) input; modes documented inline."""
    order = np.argsort(w.read_start)
    s = int(order[rng.integers(0, len(order) - 20)])
    path, steps = w.chain(s, max_len=int(rng.integers(2, 12)), flip_all=bool(rng.integers(0, 2)),
                          dense=bool(rng.integers(0, 2)))
    if len(path) < 2:
        return None
    path, steps = copy.deepcopy(path), copy.deepcopy(steps)
    rows, vm, contains = w.rows, w.vm, {}
    mode = int(rng.integers(0, 8))

    def shrink(st):  # make overlaps of the same anchor on different path edges disjoint -> several cliques
        for a in list(st["em"]):
            if rng.integers(0, 3) == 0:
                lo, hi = st["em"][a]
                st["em"][a] = (lo, lo + (hi - lo) // 4) if rng.integers(0, 2) else (hi - (hi - lo) // 4, hi)

    if mode == 1:  # arbitrary read directions, e_NONE included
        for p in path:
            p["dir"] = (True, False, None)[int(rng.integers(0, 3))]
    elif mode == 2:  # several EdgeOrders per path edge
        for st in steps:
            o = st["orders"][0]
            for _ in range(int(rng.integers(1, 3))):
                ids = list(o["ids"])
                if len(ids) > 2 and rng.integers(0, 2):
                    ids = ids[:len(ids) // 2 + 1]
                if rng.integers(0, 2):
                    ids = ids[::-1]
                st["orders"].append({"ids": ids, "score": int(o["score"]) + int(rng.integers(-5, 6)), "base": o["base"]})
    elif mode == 3:  # anchors that leave and come back -> kinks / modifiers
        for st in steps:
            for o in st["orders"]:
                if len(o["ids"]) > 2 and rng.integers(0, 2):
                    o["ids"] = [o["ids"][0], o["ids"][-1]]
    elif mode == 4:
        for st in steps:
            shrink(st)
    elif mode == 5:  # contained reads
        for p in path:
            c = w.contained_in(p["id"])
            if c:
                contains[p["id"]] = c
    elif mode >= 6:  # nested (6) / equal (7) nanopore ranges on a read
        rows = w.rows.copy()
        on_path = {p["id"] for p in path}
        idx = [i for i in range(len(rows)) if int(rows[i]["read_id"]) in on_path]
        for i in idx:
            if rng.integers(0, 3) == 0:
                j = idx[rng.integers(0, len(idx))]
                if rows[j]["read_id"] == rows[i]["read_id"]:
                    if mode == 6:
                        rows[i]["n_lo"] = rows[j]["n_lo"] + 5
                        rows[i]["n_hi"] = max(rows[j]["n_hi"] - 5, rows[i]["n_lo"] + 10)
                    else:
                        rows[i]["n_lo"], rows[i]["n_hi"] = rows[j]["n_lo"], rows[j]["n_hi"]
        vm = {(int(r["read_id"]), int(r["anchor_id"])): r for r in rows}
        if rng.integers(0, 2):
            for st in steps:
                shrink(st)
    return path, steps, rows, vm, contains


def compare(w, asm, r, path_idx=0):
    raw = apply_pieces(asm.pieces, w.flat, revcomp)
    p, qs = asm.paths[path_idx], asm.queries
    qs = qs[int(p["query_begin"]):int(p["query_end"])]
    assert raw[int(p["target_raw_off"]): int(p["target_raw_off"]) + int(p["target_len"])] == r["target"]
    assert (int(p["n_anchors"]), int(p["n_anchor_edges"])) == (r["n_anchors"], r["n_anchor_edges"])
    assert (int(p["border_lo"]), int(p["border_hi"])) == r["borders"]
    assert len(qs) == len(r["queries"])
    for q, (name, seq, lb, rb) in zip(qs, r["queries"]):
        assert raw[int(q["raw_off"]): int(q["raw_off"]) + int(q["len"])] == seq, name
        assert (int(q["lb"]), int(q["rb"])) == (lb, rb), name
        assert name.startswith(("Middle", "Left", "Right", "Contain_Illumina_Match", "Contain_Nano_Middle")[int(q["kind"])].encode())


def test_chains_match_oracle_and_paf_text(world):
    from oracle.ms_assemble_py import assemble_path
    w = world
    order = np.argsort(w.read_start)
    asm, want = Assembly(w.store), []
    # any order: the table is indexed by (read, anchor, line).  Padded past 2^16 rows so that the multi-threaded
    # install runs, with rows of foreign reads and with higher-line duplicates of real pairs (which must lose).
    rng = np.random.default_rng(3)
    pad = np.zeros(70_000, dtype=w.rows.dtype)
    pad["read_id"] = int(w.rows["read_id"].max()) + 1 + rng.integers(0, 500, len(pad))
    pad["anchor_id"] = rng.integers(0, int(w.rows["anchor_id"].max()) + 1, len(pad))
    pad["line"] = 10**6 + np.arange(len(pad))
    dup = w.rows[rng.choice(len(w.rows), 50, replace=False)].copy()
    dup["line"] += 2 * 10**6
    dup["n_lo"] += 11
    table = np.concatenate([w.rows[::-1], pad, dup])
    rng.shuffle(table)
    asm.set_rows(table)
    for k, s in enumerate(order[:60:4]):
        for flip in (False, True):
            path, steps = w.chain(int(s), max_len=8, flip_all=flip)
            if len(path) < 2:
                continue
            want.append(assemble_path(path, steps, w.vm, {}, w.nano, w.illu, k))
            asm.add_path(path, steps, None if flip else w.rows, None, k)  # flip: rows come from set_rows
    assert len(want) >= 20
    for i, r in enumerate(want):
        compare(w, asm, r, i)
    assert asm.text(2) == b"".join(r["paf"] for r in want)  # temp_1.align.paf, path order
    with pytest.raises(MsgpuError) as e:  # no device behind this context: the bases cannot be produced
        asm.finish()
    assert e.value.code == _lib.E_NODEVICE


@pytest.mark.parametrize("n_pad", [40, 70_000])
def test_row_table_in_anchor_order(world, n_pad):
    """Rows in ascending anchor-id order (what a PAF grouped by query yields): msgpu_assembly_set_rows records where
    each anchor starts instead of sorting.  With anchor ids the table never mentions (between and after the ones it
    has), higher-line duplicates ahead of the rows they repeat, one thread (40 extra rows) and many (70 k)."""
    from oracle.ms_assemble_py import assemble_path
    w = world
    rng = np.random.default_rng(11)
    top = int(w.rows["anchor_id"].max())
    pad = np.zeros(n_pad, dtype=w.rows.dtype)
    pad["read_id"] = int(w.rows["read_id"].max()) + 1 + rng.integers(0, 500, n_pad)
    pad["anchor_id"] = top + 3 + 4 * rng.integers(0, 2000, n_pad)  # gaps of unused ids
    pad["line"] = 10**6 + np.arange(n_pad)
    dup = w.rows[rng.choice(len(w.rows), 50, replace=False)].copy()
    dup["line"] += 2 * 10**6
    dup["n_lo"] += 11
    table = np.concatenate([w.rows, pad, dup])
    table = table[np.lexsort((-table["line"].astype(np.int64), table["anchor_id"]))]
    asm = Assembly(w.store)
    asm.set_rows(table)
    want = []
    order = np.argsort(w.read_start)
    for k, s in enumerate(order[:40:4]):
        path, steps = w.chain(int(s), max_len=8)
        if len(path) < 2:
            continue
        want.append(assemble_path(path, steps, w.vm, {}, w.nano, w.illu, k))
        asm.add_path(path, steps, None, None, k)
    assert len(want) >= 5
    for i, r in enumerate(want):
        compare(w, asm, r, i)
    asm.close()


def test_fuzzed_paths_match_oracle_every_branch(world):
    from oracle.ms_assemble_py import AssemblyError, assemble_path
    w = world
    rng = np.random.default_rng(5)
    cover, n_ok, n_err = {}, 0, 0
    for trial in range(500):
        case = fuzz_case(w, rng, trial)
        if case is None:
            continue
        path, steps, rows, vm, contains = case
        ocont = {k: [dict(nano=c["nano"], dir=c["dir"], matches=c["matches"]) for c in v] for k, v in contains.items()}
        pcont = {k: [dict(nano=c["nano"], dir=c["dir"], anchors=list(c["matches"])) for c in v]
                 for k, v in contains.items()}
        try:
            r = assemble_path(path, steps, vm, ocont, w.nano, w.illu, trial)
        except (AssemblyError, KeyError, IndexError):
            r = None  # the reference terminates / hangs / reads past a container on this input
        asm = Assembly(w.store)
        if r is None:
            with pytest.raises(MsgpuError) as e:
                asm.add_path(path, steps, rows, pcont, trial)
            assert e.value.code in (_lib.E_LAYOUT, _lib.E_ARG), "trial %d" % trial
            assert len(asm.paths) == 0 and asm.raw_bytes == 0  # a rejected path leaves nothing behind
            n_err += 1
            continue
        asm.add_path(path, steps, rows, pcont, trial)
        compare(w, asm, r)
        assert asm.text(2) == r["paf"]
        for k, v in r["cover"].items():
            cover[k] = cover.get(k, 0) + (v > 0)
        n_ok += 1
    assert n_ok > 250 and n_err > 20
    for k in ("multi_order", "kinks", "multi_clique", "flips", "nr_ties", "extra_groups", "dup_edges", "no_seq",
              "contain_records"):
        assert cover.get(k, 0) > 0, "branch %s never exercised: %r" % (k, cover)


def test_batch_on_threads_equals_one_by_one(world):
    """msgpu_assembly_add_paths (the assemblePaths fan-out): same records in the same order for any thread count; a
    path the reference cannot assemble is reported and skipped."""
    w = world
    order = np.argsort(w.read_start)
    cases = []
    for k, s in enumerate(order[:120:5]):
        path, steps = w.chain(int(s), max_len=9, dense=bool(k & 1))
        if len(path) >= 2:
            cases.append((path, steps))
    bad = (cases[0][0], [dict(st, orders=[]) for st in cases[0][1]])  # no EdgeOrder on its edges
    cases.insert(3, bad)
    one = Assembly(w.store)
    one.set_rows(w.rows)
    for i, (p, st) in enumerate(cases):
        if i == 3:
            with pytest.raises(MsgpuError):
                one.add_path(p, st, None, None, i)
        else:
            one.add_path(p, st, None, None, i)
    for threads in (1, 4, 64):
        many = Assembly(w.store)
        many.set_rows(w.rows)
        status = many.add_prepared_batch([Assembly.prepare(p, st, None, None, i) for i, (p, st) in enumerate(cases)],
                                         threads)
        assert list(np.nonzero(status)[0]) == [3] and status[3] == _lib.E_LAYOUT
        assert many.text(2) == one.text(2)
        assert many.pieces.tobytes() == one.pieces.tobytes()
        assert many.paths.tobytes() == one.paths.tobytes() and many.queries.tobytes() == one.queries.tobytes()


def test_contig_of_exact_reads_is_the_genome_up_to_joint_duplicates(oracle, world):
    """Size-independent property: with error-free reads and exact PAF coordinates the contig is the genome span of the
    path except for the bases the reference's inclusive slices duplicate at every joint (SequenceUtils.cpp:27-38)."""
    from oracle.ms_assemble_py import assemble_path
    w = World(300, 5000, 1500, 7, jitter=0)
    w.attach(oracle.overlap(w.rows))
    order = np.argsort(w.read_start)
    for flip in (False, True):
        path, steps = w.chain(int(order[3]), max_len=10, flip_all=flip)
        r = assemble_path(path, steps, w.vm, {}, w.nano, w.illu, 1)
        lo = min(w.read_start[p["id"]] for p in path)
        hi = max(w.read_start[p["id"]] for p in path) + w.read_len
        t = revcomp(r["target"]) if flip else r["target"]
        d = oracle.edit_distance_banded(t, w.genome[lo:hi], 400)
        assert d <= 6 * r["n_anchors"], (d, r["n_anchors"])  # ~4 duplicated bases per placed anchor
        assert abs(len(t) - (hi - lo)) == d  # pure insertions


def test_two_reads_one_anchor_by_hand(world):
    """Smallest path: both reads carry one common anchor -> one anchor vertex, no DAG edge; the contig is
    Left(longest) + anchor + Right(longest) (ap.cpp:886-895, 1012-1032)."""
    from oracle.ms_assemble_py import assemble_path
    from oracle.ms_oracle_py import (get_anchor_sequence, get_sequence_left_of_anchor, get_sequence_right_of_anchor,
                                     str_slice)
    w = world
    order = np.argsort(w.read_start)
    path, steps = w.chain(int(order[5]), max_len=2)
    a = steps[0]["orders"][0]["ids"][0]
    steps = [{"orders": [dict(steps[0]["orders"][0], ids=[a])], "em": steps[0]["em"]}]
    r = assemble_path(path, steps, w.vm, {}, w.nano, w.illu, 3)
    ov = steps[0]["em"][a]
    anchor = get_anchor_sequence(w.vm[(path[0]["id"], a)], w.illu[a], ov, path[0]["dir"])
    lefts = [get_sequence_left_of_anchor(w.vm[(p["id"], a)], w.nano[p["id"]], w.illu[a], p["len"], ov, p["dir"])
             for p in path]
    rights = [get_sequence_right_of_anchor(w.vm[(p["id"], a)], w.nano[p["id"]], w.illu[a], p["len"], ov, p["dir"])
              for p in path]
    left, right = max(lefts, key=len), max(rights, key=len)
    # updateConsensusBase: strSlice(new, 0, old_lo - new_lo) is inclusive, so one base more than the gap is taken
    want = str_slice(left, 0, len(left)) + anchor + str_slice(right, -len(right), len(right))
    assert r["target"] == want
    assert r["n_anchors"] == 1 and r["n_anchor_edges"] == 0
    asm = Assembly(w.store)
    asm.add_path(path, steps, w.rows, None, 3)
    compare(w, asm, r)


def test_limit_length_and_headers():
    from oracle.ms_assemble_py import limit_length
    assert limit_length(b"") == b""
    assert limit_length(b"A" * 60) == b"A" * 60
    assert limit_length(b"A" * 61) == b"A" * 60 + b"\n" + b"A"
    assert limit_length(b"A" * 120) == b"A" * 60 + b"\n" + b"A" * 60
    L = _lib.lib()
    for n in (0, 1, 59, 60, 61, 120, 121, 6001):
        assert L.msgpu_fasta_text_bytes(5, n) == 5 + len(limit_length(b"A" * n)) + 1


def test_row_table_install_at_scale(oracle, tmp_path):
    """msgpu_assembly_set_rows on a table large enough for its threaded two-level counting sort (cfg2: 512 k rows),
    grouped by anchor and shuffled with duplicate (read, anchor) rows of higher line numbers: every layout that looks
    its VertexMatches up in the installed table equals the layout that was handed exactly its own rows."""
    from muchsalsa_amd import synth
    cfg = synth.CONFIGS["cfg2"]
    rows, read_names, anchor_names = synth.accepted_rows(synth.paf_table(**cfg))
    tables = oracle.overlap(rows)
    G, r_start, r_fwd = synth.read_layout(cfg["n_reads"], cfg["read_len"], cfg["seed"])
    ro = np.array([int(n[1:]) for n in read_names])
    paths = synth.chain_paths(tables, r_start[ro], r_fwd[ro], cfg["read_len"], G // 4, max_reads=10, max_paths=12)
    assert len(paths) >= 8
    # layout only: no base is read, but every slice is clipped to its sequence's length
    _, a_len = synth.anchor_layout(cfg["n_reads"], cfg["read_len"], cfg["n_anchors"], cfg["seed"])
    ao = np.array([int(n[1:]) for n in anchor_names])
    with open(tmp_path / "n.fa", "wb") as f:
        f.write(b"".join(b">r%d\n" % i + b"A" * cfg["read_len"] + b"\n" for i in range(len(read_names))))
    with open(tmp_path / "i.fa", "wb") as f:
        f.write(b"".join(b">u%d\n" % i + b"C" * int(a_len[o]) + b"\n" for i, o in enumerate(ao)))
    store = SeqStore(device=-1)
    store.upload(NANOPORE, SeqFile(str(tmp_path / "n.fa")))
    store.upload(ILLUMINA, SeqFile(str(tmp_path / "i.fa")))
    ref = Assembly(store)
    for k, (p, st) in enumerate(paths):  # every path with exactly the rows of its own reads
        mine = rows[np.isin(rows["read_id"], [r["id"] for r in p])]
        ref.add_path(p, st, mine, None, k)
    rng = np.random.default_rng(3)
    dup = rows[rng.choice(len(rows), 5000, replace=False)].copy()
    dup["line"] = rows["line"].max() + 1 + np.arange(len(dup))  # the lowest line must win (MatchMap.cpp:64-80)
    dup["n_lo"] += 11
    shuffled = np.concatenate([rows, dup])
    rng.shuffle(shuffled)
    # ascending anchor ids with the duplicates INSIDE their scaffolds and ahead of the rows they repeat (the table that
    # needs no sort: msgpu_assembly_set_rows only records where every anchor starts)
    both = np.concatenate([rows, dup])
    by_anchor = both[np.lexsort((-both["line"].astype(np.int64), both["anchor_id"]))]
    by_read = both[np.argsort(both["read_id"], kind="stable")]
    assert np.all(np.diff(rows["anchor_id"].astype(np.int64)) >= 0)
    for name, table in (("grouped by anchor", rows), ("shuffled + duplicates", shuffled),
                        ("ascending anchors + duplicates", by_anchor), ("grouped by read + duplicates", by_read)):
        for copy in (True, False):  # copied, or read where the caller keeps it (msgpu_assembly_borrow_rows)
            got = Assembly(store)
            got.set_rows(table, copy=copy)
            status = got.add_prepared_batch([Assembly.prepare(p, st, None, None, k) for k, (p, st) in enumerate(paths)], 4)
            assert not status.any(), name
            assert got.text(2) == ref.text(2), name
            assert got.pieces.tobytes() == ref.pieces.tobytes(), name
            got.close()
    ref.close()
    store.close()
