"""Sequence side of the consensus stage (A9): the oracle against the REFERENCE'S OWN fixtures
(libms/tests/SA_test.cpp golden strings + test_data files, copied as data into tests/golden/ref_test_data)."""
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_test_data")


@pytest.fixture(scope="module")
def sa():
    return json.load(open(os.path.join(GOLD, "sa_test_expected.json")))


def test_fasta_whole_record_fetch_matches_SA_test(oracle, sa):
    # SATest.FastaTest (SA_test.cpp:11-66): fasta.fa used as both the nanopore and the illumina file
    names, seqs = oracle.seq_load(os.path.join(GOLD, "fasta.fa"))
    assert names == sa["FastaTest"]["names"]
    assert [s.decode() for s in seqs] == sa["FastaTest"]["sequences"]
    assert [len(s) for s in seqs] == [1231, 1020]


def test_fastq_whole_record_fetch_matches_SA_test(oracle, sa):
    # SATest.FastQTest (SA_test.cpp:68-136): ids cut at the first whitespace; '+' and quality lines skipped
    names, seqs = oracle.seq_load(os.path.join(GOLD, "fastq.fq"))
    assert names == ["A00456:495:HHVKWDSXY:1:1101:25952:1031", "A00456:495:HHVKWDSXY:1:1101:3016:1047"]
    assert [s.decode() for s in seqs] == sa["FastQTest"]["nanopore"]
    _, illu = oracle.seq_load(os.path.join(GOLD, "fasta.fa"))
    assert [s.decode() for s in illu] == sa["FastQTest"]["illumina"]


def py_str_slice(s, i, j):
    """strSlice (SequenceUtils.cpp:27-38) written out: Python-like indices but an INCLUSIVE, clipped end."""
    size = len(s)
    i2 = i if i >= 0 else size + i
    j2 = j if j >= 0 else size + j
    start = max(0, i2)
    end = max(min(size, max(0, j2)), i2 % (1 << 64))
    if start > size:
        return None  # std::out_of_range
    return s[start:start + (end - start + 1)]


def test_str_slice_semantics(oracle):
    s = b"ABCDEFGHIJ"
    cases = [(0, 3), (2, 2), (0, 9), (0, 10), (0, 99), (5, 3), (-3, -1), (-3, 9), (3, -2), (0, 0), (9, 9), (10, 12),
             (-20, 2), (4, -20)]
    for i, j in cases:
        off, n = oracle.str_slice(len(s), i, j)
        want = py_str_slice(s, i, j)
        assert s[off:off + n] == (want if want is not None else b""), (i, j)
    # the quirk the survey points out: getXSequence(l, r) yields r - l + 2 characters when in range
    assert oracle.get_sequence(s, 2, 4, True) == b"CDEF"
    assert oracle.get_sequence(s, 2, 4, False) == b"FEDC"[::1].translate(bytes.maketrans(b"ACGT", b"TGCA"))


def test_reverse_complement_only_maps_upper_case_acgt(oracle):
    s = b"ACGTNacgtRYKM-*"
    got = oracle.get_sequence(s, 0, len(s), False)
    assert got == s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    assert got == b"*-MKYRtgcaNACGT"
