"""Row tables with reads of mixed length, so that short reads lie inside long ones: contained + primary EdgeOrders, the
input of the graph clean-up after the overlap path (findContractionEdges / sanityCheck, src/main.cpp:416-463)."""
import numpy as np

from muchsalsa_amd.synth import ROW_DTYPE


def varlen_rows(n_reads, n_anchors, genome, seed, len_lo=1500, len_hi=9000, jitter=10, tiled=False, layout=None):
    """tiled: anchors are consecutive, non-overlapping stretches of the genome separated by gaps of 0..300 bases (what
    unitigs of one genome look like) instead of independent random intervals; n_anchors is then ignored.
    layout (optional dict) receives r_start, r_len, r_fwd, a_start, a_len indexed by Registry id."""
    rng = np.random.default_rng(seed)
    r_len = rng.integers(len_lo, len_hi + 1, n_reads)
    r_start = rng.integers(0, genome - r_len)
    r_fwd = rng.integers(0, 2, n_reads).astype(bool)
    if tiled:
        starts, lens, pos = [], [], int(rng.integers(0, 200))
        while pos + 1500 < genome:
            ln = int(rng.integers(500, 1501))
            starts.append(pos)
            lens.append(ln)
            pos += ln + int(rng.integers(0, 301))
        a_start, a_len = np.array(starts), np.array(lens)
        n_anchors = len(starts)
    else:
        a_len = rng.integers(500, 1501, n_anchors)
        a_start = rng.integers(0, genome - a_len)
    recs = []
    for a in range(n_anchors):  # rows grouped by anchor, then read: what a PAF sorted by query looks like
        lo = np.maximum(a_start[a], r_start)
        hi = np.minimum(a_start[a] + a_len[a], r_start + r_len)
        for r in np.nonzero(hi - lo >= 420)[0]:
            q_lo, q_hi = lo[r] - a_start[a], hi[r] - a_start[a]
            if r_fwd[r]:
                t_lo, t_hi = lo[r] - r_start[r], hi[r] - r_start[r]
            else:
                t_lo, t_hi = r_start[r] + r_len[r] - hi[r], r_start[r] + r_len[r] - lo[r]
            t_lo = max(0, int(t_lo) + int(rng.integers(-jitter, jitter + 1)))
            t_hi = min(int(r_len[r]), int(t_hi) + int(rng.integers(-jitter, jitter + 1)))
            nm = int((q_hi - q_lo) * rng.uniform(0.86, 0.97))
            if nm < 400 or q_hi - q_lo < 400:
                continue
            prim = (q_hi - q_lo) >= 500 and nm >= 500
            recs.append((a, r, int(r_len[r]), int(q_lo), int(q_hi) - 1, t_lo, t_hi - 1, nm, len(recs),
                         int(r_fwd[r]) | (int(prim) << 1)))
    rows = np.array(recs, dtype=ROW_DTYPE)
    orig = {}
    for f in ("read_id", "anchor_id"):  # Registry ids: first-seen order (Registry.cpp:36-45)
        uniq, first, inv = np.unique(rows[f], return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        by_first = np.argsort(first, kind="stable")
        rank[by_first] = np.arange(len(uniq))
        rows[f] = rank[inv]
        orig[f] = uniq[by_first]  # Registry id -> generator index
    if layout is not None:
        r, a = orig["read_id"], orig["anchor_id"]
        layout.update(r_start=r_start[r], r_len=r_len[r], r_fwd=r_fwd[r], a_start=a_start[a], a_len=a_len[a],
                      genome=genome)
    return rows


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def make_dataset(d, seed, jitter, fastq, n_reads=400, genome_len=250_000):
    """Write contigs.paf, unitigs.fa and nanopore.fa/.fq for a tiled-anchor data set into directory d (a pathlib.Path)
    -> (rows, layout, genome, nano {id: bytes}, illu {id: bytes}, name of the read file)"""
    lay = {}
    rows = varlen_rows(n_reads, 0, genome_len, seed, tiled=True, layout=lay, jitter=jitter)
    genome = np.random.default_rng(99 + seed).choice(np.frombuffer(b"ACGT", dtype=np.uint8), genome_len).tobytes()
    nano, illu = {}, {}
    for i in range(len(lay["r_start"])):
        s = genome[int(lay["r_start"][i]): int(lay["r_start"][i]) + int(lay["r_len"][i])]
        nano[i] = s if lay["r_fwd"][i] else s.translate(_COMP)[::-1]
    for j in range(len(lay["a_start"])):
        illu[j] = genome[int(lay["a_start"][j]): int(lay["a_start"][j]) + int(lay["a_len"][j])]
    with open(d / "contigs.paf", "w") as f:  # one line per row, in line order, + the line the reference never parses
        for r in rows:
            a, rd = int(r["anchor_id"]), int(r["read_id"])
            f.write("u%d\t%d\t%d\t%d\t%s\tr%d\t%d\t%d\t%d\t%d\t%d\t60\n" % (
                a, len(illu[a]), r["i_lo"], int(r["i_hi"]) + 1, "+" if int(r["flags"]) & 1 else "-", rd, r["read_len"],
                r["n_lo"], int(r["n_hi"]) + 1, r["score"], int(r["i_hi"]) + 1 - int(r["i_lo"])))
        f.write("u0\t1\t0\t1\t+\tr0\t1\t0\t1\t0\t1\t0\n")
    with open(d / "unitigs.fa", "wb") as f:
        for j in sorted(illu, reverse=True):  # file order is unrelated to Registry order
            f.write(b">u%d some description\n" % j)
            for k in range(0, len(illu[j]), 70):
                f.write(illu[j][k:k + 70] + b"\n")
    name = "nanopore.fq" if fastq else "nanopore.fa"
    with open(d / name, "wb") as f:
        for i in sorted(nano):
            if fastq:
                f.write(b"@r%d\n" % i + nano[i] + b"\n+\n" + b"I" * len(nano[i]) + b"\n")
            else:
                f.write(b">r%d\n" % i + nano[i] + b"\n")
    return rows, lay, genome, nano, illu, name


def synth_sequences(shape, read_names, anchor_names):
    """The reads / unitigs of a muchsalsa_amd.synth workload (CONFIGS or TILED entry) keyed by Registry id, as the bytes
    the FASTA files hold -> (genome length, read starts, read strands, read lengths, nano {id: bytes}, illu {id: bytes})"""
    from muchsalsa_amd import synth
    n_reads, L, seed = shape["n_reads"], shape["read_len"], shape["seed"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, seed, read_len_min=shape.get("read_len_min"))
    r_len = synth.read_lengths(n_reads, L, seed, shape.get("read_len_min"))
    a_start, a_len = synth.anchor_layout(n_reads, L, shape["n_anchors"], seed, tiled=shape.get("tiled", False))
    genome = synth.genome_bases(G, seed).tobytes()
    ro = np.array([int(n[1:]) for n in read_names])
    ao = np.array([int(n[1:]) for n in anchor_names])
    nano = {}
    for i, o in enumerate(ro):
        s = genome[r_start[o]: r_start[o] + r_len[o]]
        nano[i] = s if r_fwd[o] else s.translate(_COMP)[::-1]
    illu = {i: genome[a_start[o]: a_start[o] + a_len[o]] for i, o in enumerate(ao)}
    return G, r_start[ro], r_fwd[ro], r_len[ro], nano, illu
