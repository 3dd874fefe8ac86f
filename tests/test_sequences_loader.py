"""msgpu_seq_parse / msgpu_str_slice (host code of libmsgpu; no GPU needed) against the reference's SA_test fixtures
and against the oracle."""
import json
import os

import numpy as np
import pytest

from muchsalsa_amd import sequences

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_test_data")


@pytest.fixture(scope="module")
def sa():
    return json.load(open(os.path.join(GOLD, "sa_test_expected.json")))


def test_fasta_matches_SA_test(sa):
    f = sequences.SeqFile(os.path.join(GOLD, "fasta.fa"))
    assert f.names == sa["FastaTest"]["names"]
    assert [f.sequence(i).decode() for i in range(len(f))] == sa["FastaTest"]["sequences"]


def test_fastq_matches_SA_test(sa):
    f = sequences.SeqFile(os.path.join(GOLD, "fastq.fq"))
    assert [f.sequence(i).decode() for i in range(len(f))] == sa["FastQTest"]["nanopore"]
    assert f.names == ["A00456:495:HHVKWDSXY:1:1101:25952:1031", "A00456:495:HHVKWDSXY:1:1101:3016:1047"]


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_bytes(text)
    return str(p)


@pytest.mark.parametrize("name,text", [
    ("a.fa", b">r1 desc\nACGT\nNN\n>r2\n\nTT TT\r\n>r1\nGGGG\n>r3"),           # dup id (first wins), blank/CRLF/space, empty
    ("b.fasta", b"junk before\n>x\tdesc\nAC\nGT"),                                # no trailing newline, tab in header
    ("c.fq", b"@q1 a\nACGT\nAC\n+\nFFFF\nFF\n@q2\nTTTT\n+q2\nIIII\n"),          # multi-line FASTQ record
    ("d.txt", b"@only\nACGT\n+\n!!!!\n"),                                         # unknown extension -> FASTQ
    ("e.fa", b""), ("f.fa", b"no records here\n"),
])
def test_loader_matches_oracle_on_edge_cases(oracle, tmp_path, name, text):
    path = _write(tmp_path, name, text)
    want_names, want_seqs = oracle.seq_load(path)
    f = sequences.SeqFile(path)
    assert f.names == want_names
    assert [f.sequence(i) for i in range(len(f))] == want_seqs


def test_first_cases_by_hand(tmp_path):
    f = sequences.SeqFile(_write(tmp_path, "a.fa", b">r1 desc\nACGT\nNN\n>r2\n\nTT TT\r\n>r1\nGGGG\n>r3"))
    assert f.names == ["r1", "r2", "r3"]
    assert [f.sequence(i) for i in range(3)] == [b"ACGTNN", b"TTTT", b""]


def test_str_slice_matches_oracle(oracle):
    rng = np.random.default_rng(1)
    for _ in range(3000):
        size = int(rng.integers(0, 40))
        i, j = (int(x) for x in rng.integers(-60, 60, 2))
        assert sequences.str_slice(size, i, j) == oracle.str_slice(size, i, j), (size, i, j)


def _records(f):
    return f.names, [f.sequence(i) for i in range(len(f))]


@pytest.mark.parametrize("threads", [2, 3, 7, 16])
def test_chunked_parse_equals_sequential(oracle, tmp_path, monkeypatch, threads):
    """Large files are parsed in chunks on several threads, each chunk by the same sequential state machine, the cuts
    verified afterwards (MSGPU_SEQ_THREADS forces the chunked path at any size).  Same records as one pass -- and as the
    oracle -- for multi-line FASTA, duplicates across chunks (first wins), FASTQ, and a FASTQ whose quality lines begin
    with '@' (which '@' lines are descriptions depends on everything before them: the cuts do not verify and the file
    takes the sequential pass)."""
    rng = np.random.default_rng(threads)
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)

    def seq(n):
        return bases[rng.integers(0, 5, n)].tobytes()
    fa = b"leading junk\n"
    for i in range(400):
        s = seq(int(rng.integers(0, 900)))
        name = b"r%d" % (i if i % 37 else i // 2)  # every 37th record repeats an earlier id
        fa += b">" + name + b" some description\n" + b"\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + b"\n"
    fq = b""
    for i in range(300):
        s = seq(int(rng.integers(1, 300)))
        fq += b"@q%d x\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n"
    evil = b""
    for i in range(300):  # quality lines that start with '@', sequence lengths that vary: cuts land on them
        s = seq(int(rng.integers(1, 200)))
        evil += b"@e%d\n" % i + s + b"\n+\n" + b"@" + b"F" * (len(s) - 1) + b"\n"
    evil3 = b""
    for i in range(300):  # three-line records whose first quality line begins with '@' and whose third begins with '+'
        s = seq(int(rng.integers(3, 200)))
        a, b = len(s) // 3, 2 * len(s) // 3
        q = b"@" + b"F" * (a - 1), b"F" * (b - a), b"+" + b"F" * (len(s) - b - 1)
        evil3 += b"@t%d\n" % i + s[:a] + b"\n" + s[a:b] + b"\n" + s[b:] + b"\n+\n" + b"\n".join(q) + b"\n"
    for name, text in (("m.fa", fa), ("s.fq", fq), ("evil.fq", evil), ("evil3.fq", evil3), ("one.fa", b">x\nACGT\n"), ("none.fa", b"")):
        path = _write(tmp_path, name, text)
        monkeypatch.setenv("MSGPU_SEQ_THREADS", "1")
        want = _records(sequences.SeqFile(path))
        monkeypatch.setenv("MSGPU_SEQ_THREADS", str(threads))
        f = sequences.SeqFile(path)
        got = _records(f)
        assert got == want, (name, threads)
        # what goes to HBM: the records where they lie in ONE buffer, nothing but bases between them (the 2-bit form of the
        # store keeps a list of every other byte)
        buf = f.buffer()
        assert len(buf) <= len(text) and (name.startswith("evil") or set(buf) <= set(b"ACGTN")), (name, threads)
        o_names, o_seqs = oracle.seq_load(path)
        assert got == (o_names, o_seqs), (name, "oracle")
    assert len(_records(sequences.SeqFile(_write(tmp_path, "m2.fa", fa)))[0]) == len({b"r%d" % (i if i % 37 else i // 2) for i in range(400)})
