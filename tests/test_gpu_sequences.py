"""GPU parity of the sequence store + gather kernel (device half of the consensus stage) through the C-ABI:
against the reference's own SA_test fixtures, against the oracle's strSlice/reverse-complement, and against a
host statement of updateConsensusBase (ap.cpp:205-229)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_test_data")


@pytest.fixture(autouse=True, params=["bytes", "two_bit"])
def store_form(request, monkeypatch):
    """every test of this module runs on the byte-per-base store and on the 2-bit store (msgpu_seq_pack)"""
    if request.param == "two_bit":
        from muchsalsa_amd import sequences as S
        plain_plan = S.SeqStore.plan

        def plan(self, pieces):
            self.pack()  # idempotent; both stores are uploaded by the time a test plans its first gather
            return plain_plan(self, pieces)
        monkeypatch.setattr(S.SeqStore, "plan", plan)
    return request.param


def _run(store, pieces, total=None):
    import torch
    plan = store.plan(pieces)
    n = store.plan_out_bytes(plan) if total is None else total
    out = torch.full((max(n, 1) + 64,), 0x2e, dtype=torch.uint8, device="cuda:0")  # '.' canary
    torch.cuda.synchronize()  # the store runs on its own non-blocking stream: finish the fill first
    store.run(plan, out.data_ptr(), n)
    store.synchronize()
    host = out.cpu().numpy()
    assert (host[n:] == 0x2e).all(), "wrote past the end of the output"
    return host[:n].tobytes()


def _write_fasta(path, seqs, prefix, width=60):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">%s%d extra words\n" % (prefix, i))
            for k in range(0, len(s), width):
                f.write(s[k:k + width] + "\n")


def test_whole_records_match_the_reference_SA_test_fixtures():
    from muchsalsa_amd import sequences as S
    sa = json.load(open(os.path.join(GOLD, "sa_test_expected.json")))
    with S.SeqStore(0) as store:
        fa, fq = S.SeqFile(os.path.join(GOLD, "fasta.fa")), S.SeqFile(os.path.join(GOLD, "fastq.fq"))
        store.upload(S.ILLUMINA, fa)
        store.upload(S.NANOPORE, fq)
        for kind, want in ((S.ILLUMINA, sa["FastQTest"]["illumina"]), (S.NANOPORE, sa["FastQTest"]["nanopore"])):
            off = 0
            pieces = []
            for i, w in enumerate(want):
                # getXSequence(id, 0, size, true) = strSlice(seq, 0, size + 1) = the whole record
                pieces.append(store.resolve(kind, i, 0, len(w), True, dst_off=off))
                off += len(w)
            assert _run(store, np.array(pieces)).decode() == "".join(want)


def test_random_slices_match_oracle(oracle, tmp_path):
    from muchsalsa_amd import sequences as S
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGTACGTACGTNacgtRY", dtype=np.uint8)
    seqs = ["".join(map(chr, rng.choice(alphabet, int(n)))) for n in rng.integers(1, 3000, 40)]
    unis = ["".join(map(chr, rng.choice(alphabet, int(n)))) for n in rng.integers(1, 1500, 25)]
    _write_fasta(tmp_path / "reads.fa", seqs, "r")
    _write_fasta(tmp_path / "unitigs.fa", unis, "u")
    with S.SeqStore(0) as store:
        fr, fu = S.SeqFile(str(tmp_path / "reads.fa")), S.SeqFile(str(tmp_path / "unitigs.fa"))
        assert [fr.sequence(i).decode() for i in range(len(fr))] == seqs
        # Registry ids need not follow file order
        rid = rng.permutation(len(seqs)).astype(np.uint32)
        store.upload(S.NANOPORE, fr, rid, len(seqs))
        store.upload(S.ILLUMINA, fu)
        by_id = {int(rid[i]): seqs[i] for i in range(len(seqs))}
        pieces, want, off = [], [], 0
        for _ in range(4000):
            kind = int(rng.integers(0, 2))
            pool = by_id if kind == S.NANOPORE else dict(enumerate(unis))
            sid = int(rng.integers(0, len(pool)))
            s = pool[sid].encode()
            left = int(rng.integers(-20, len(s) + 20))
            right = int(rng.integers(-20, len(s) + 40))
            direction = bool(rng.integers(0, 2))
            w = oracle.get_sequence(s, left, right, direction)
            p = store.resolve(kind, sid, left, right, direction, dst_off=off)
            assert int(p["len"]) == len(w), (len(s), left, right)
            pieces.append(p)
            want.append(w)
            off += len(w)
        got = _run(store, np.array(pieces))
        assert got == b"".join(want)


def update_consensus_base(old, old_b, new, new_b):
    """updateConsensusBase, ap.cpp:205-229, on byte strings (strSlice = inclusive end)."""
    if old is None:
        return new, new_b[0], new_b[1]

    def str_slice(s, i, j):
        size = len(s)
        i2 = i if i >= 0 else size + i
        j2 = j if j >= 0 else size + j
        st = max(0, i2)
        en = max(min(size, max(0, j2)), i2)
        return s[st:st + (en - st + 1)]
    if new_b[0] < old_b[0]:
        upd = str_slice(new, 0, old_b[0] - new_b[0]) + old
    elif new_b[1] > old_b[1]:
        upd = old + str_slice(new, -(new_b[1] - old_b[1]), len(new))
    else:
        upd = old
    return upd, min(old_b[0], new_b[0]), max(old_b[1], new_b[1])


def test_stitching_like_updateConsensusBase(oracle, tmp_path):
    """A tiling of reads over a genome, stitched in a random order by the reference's prepend/append rule:
    libmsgpu's ConsensusBase (updateConsensusBase on pieces) + one gather launch == the string version."""
    from muchsalsa_amd import sequences as S
    rng = np.random.default_rng(11)
    G = 60000
    genome = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), G))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    pos = 0
    while pos < G - 3000:
        L = int(rng.integers(2000, 6000))
        fwd = bool(rng.integers(0, 2))
        seg = genome[pos:pos + L]
        reads.append((pos, min(pos + L, G), fwd, seg if fwd else seg.translate(comp)[::-1]))
        pos += int(rng.integers(500, L - 400))
    order = rng.permutation(len(reads))  # random order: exercises both prepend and append
    _write_fasta(tmp_path / "r.fa", [r[3].decode() for r in reads], "r")
    with S.SeqStore(0) as store:
        store.upload(S.NANOPORE, S.SeqFile(str(tmp_path / "r.fa")))
        cb = S.ConsensusBase()
        want, wb = None, (0, 0)
        for k in order:
            a, b, fwd, seq = reads[k]
            oriented = oracle.get_sequence(seq, 0, len(seq), fwd)  # the whole read in genome orientation
            assert oriented == genome[a:b]
            piece = store.resolve(S.NANOPORE, int(k), 0, len(seq), fwd)
            cb.update(np.array([piece]), a, b - 1)
            want, lo, hi = update_consensus_base(want, wb, oriented, (a, b - 1))
            wb = (lo, hi)
        assert cb.borders == (wb[0], wb[1], len(want))
        got = _run(store, cb.pieces())
        assert got == want
        cb.close()


def test_unaligned_tiny_and_empty_pieces(oracle, tmp_path):
    from muchsalsa_amd import sequences as S
    rng = np.random.default_rng(5)
    s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 5000))
    _write_fasta(tmp_path / "one.fa", [s.decode()], "x")
    with S.SeqStore(0) as store:
        store.upload(S.NANOPORE, S.SeqFile(str(tmp_path / "one.fa")))
        pieces, want, off = [], [], 0
        for length in list(range(0, 40)) + [63, 64, 65, 4095, 4096, 4097]:
            for start in (0, 1, 2, 3, 5, 17):
                for direction in (True, False):
                    left, right = start, start + length - 2
                    w = oracle.get_sequence(s, left, right, direction) if length else b""
                    p = store.resolve(S.NANOPORE, 0, left, right, direction, dst_off=off)
                    if length == 0:
                        p["len"] = 0
                    assert int(p["len"]) == len(w)
                    pieces.append(p)
                    want.append(w)
                    off += len(w)
        assert _run(store, np.array(pieces)) == b"".join(want)
        assert _run(store, np.zeros(0, dtype=S.COPY_DTYPE), total=0) == b""


def test_segment_builders_through_the_gather_kernel(oracle, tmp_path):
    """getAnchorSequence / getSequence{Left,Right}OfAnchor / getSequenceBetweenAnchors (ap.cpp:352-579): libmsgpu's
    piece composers + one gather launch for a whole batch of segments == the oracle's strings."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import segcases as SC
    from muchsalsa_amd import sequences as S
    rng, reads, unis = SC.make_world(21, n_reads=30, n_unitigs=40)
    _write_fasta(tmp_path / "r.fa", [r.decode() for r in reads], "r")
    _write_fasta(tmp_path / "u.fa", [u.decode() for u in unis], "u")
    with S.SeqStore(0) as store:
        store.upload(S.NANOPORE, S.SeqFile(str(tmp_path / "r.fa")))
        store.upload(S.ILLUMINA, S.SeqFile(str(tmp_path / "u.fa")))
        pieces, want, off = [], [], 0

        def emit(ps, s):
            nonlocal off
            ps = ps.copy()
            ps["dst_off"] += off
            pieces.append(ps)
            want.append(s)
            off += len(s)
        for _ in range(1500):
            m, ov = SC.random_match(rng, reads, unis)
            nano, illu = reads[int(m["read_id"])], unis[int(m["anchor_id"])]
            d = bool(rng.integers(0, 2))
            emit(store.seg_anchor(m, ov, d)[0], oracle.anchor_sequence(m, illu, ov, d))
            emit(store.seg_left_of_anchor(m, len(nano), ov, d)[0], oracle.left_of_anchor(m, nano, illu, len(nano), ov, d))
            emit(store.seg_right_of_anchor(m, len(nano), ov, d)[0],
                 oracle.right_of_anchor(m, nano, illu, len(nano), ov, d))
            m2, ov2 = SC.random_match(rng, reads, unis, read_id=int(m["read_id"]))
            ps, dist, has = store.seg_between_anchors(m, m2, ov, ov2, d)
            odist, oseq = oracle.between_anchors(m, m2, nano, illu, unis[int(m2["anchor_id"])], ov, ov2, d)
            assert dist == odist and has == (oseq is not None)
            if has:
                emit(ps, oseq)
        got = _run(store, np.concatenate(pieces))
        assert got == b"".join(want)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_parse_upload_equals_parse_then_upload(tmp_path, monkeypatch, threads):
    """msgpu_seq_parse_upload strips the records into a ring of page-locked slots that travel to HBM while the file is
    read (any number of parser threads; records far longer than a slot; duplicates; a FASTQ whose cuts do not verify and
    is parsed a second time in one pass): same names and lengths as msgpu_seq_parse, and every record read back from the
    store -- byte-per-base or 2-bit -- is the host parser's record."""
    from muchsalsa_amd import sequences as S
    rng = np.random.default_rng(threads)
    alphabet = np.frombuffer(b"ACGTACGTACGTNacgt", dtype=np.uint8)

    def seq(n):
        return alphabet[rng.integers(0, len(alphabet), int(n))].tobytes()
    fa = b"junk\n"
    for i in range(60):
        s = seq(rng.choice([0, 7, 900, 70_000, 1_300_000]) if i % 9 else 5_000_000)  # 5 MB > one 4 MiB slot
        name = b"r%d" % (i if i % 13 else i // 2)  # some ids come twice: the first record wins
        fa += b">" + name + b" x\n" + (b"\n".join(s[k:k + 20_011] for k in range(0, len(s), 20_011)) if i % 2 else s) + b"\n"
    evil = b""
    for i in range(400):  # quality lines that start with '@': the cuts land on them and the file takes the one-pass route
        s = seq(rng.integers(1, 3000))
        evil += b"@e%d\n" % i + s + b"\n+\n@" + b"F" * (len(s) - 1) + b"\n"
    evil3 = b""
    for i in range(400):  # ... and three-line records whose first quality line begins with '@' and whose third begins with '+'
        s = seq(rng.integers(3, 3000))
        a, b = len(s) // 3, 2 * len(s) // 3
        q = b"@" + b"F" * (a - 1), b"F" * (b - a), b"+" + b"F" * (len(s) - b - 1)
        evil3 += b"@t%d\n" % i + s[:a] + b"\n" + s[a:b] + b"\n" + s[b:] + b"\n+\n" + b"\n".join(q) + b"\n"
    monkeypatch.setenv("MSGPU_SEQ_THREADS", str(threads))
    small = [("a.fa", b">r1 desc\nACGT\nNN\n>r2\n\nTT TT\r\n>r1\nGGGG\n>r3"),  # dup id, blank / CRLF / space, empty record
             ("b.fasta", b"junk before\n>x\tdesc\nAC\nGT"),                     # no trailing newline, tab in the header
             ("c.fq", b"@q1 a\nACGT\nAC\n+\nFFFF\nFF\n@q2\nTTTT\n+q2\nIIII\n"),  # multi-line FASTQ record
             ("f.fa", b"no records here\n")]
    for name, text in [("big.fa", fa), ("evil.fq", evil), ("evil3.fq", evil3), ("none.fa", b"")] + small:
        path = tmp_path / name
        path.write_bytes(text)
        host = S.SeqFile(str(path))
        with S.SeqStore(0) as store:
            dev = store.parse_upload(S.NANOPORE, str(path))
            assert dev.names == host.names and [dev.length(i) for i in range(len(dev))] == [host.length(i) for i in range(len(host))]
            assert dev.buffer() == b""  # (no bytes on the host)
            store.set_ids(S.NANOPORE, dev)
            if not len(host):
                continue
            pieces, off = [], 0
            for i in range(len(host)):
                n = host.length(i)
                if n:
                    pieces.append(store.resolve(S.NANOPORE, i, 0, n, True, dst_off=off))
                    off += n
            got = _run(store, np.array(pieces))
            assert got == b"".join(host.sequence(i) for i in range(len(host))), (name, threads)
