"""BASELINE.json configs[0]: the reference's own test_data (fasta.fa as unitig AND nanopore file, exactly like
libms/tests/SA_test.cpp:24-27; fastq.fq for the FASTQ index path) plumbed through the whole CPU-visible path with a
hand-made PAF: loader -> row table -> overlap tables (oracle here; tests/test_gpu_parity.py runs the same rows on the
GPU) -> sequence store layout -> anchor segment.  test_data/ holds no PAF, so the four lines below are synthesized
(SURVEY.md section 8(d), cfg1)."""
import json
import os

import numpy as np

import ms_oracle_py as P
import segcases as SC
from muchsalsa_amd import overlap, sequences as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_test_data")

# unitig HSBGPG (1231 bp) anchors both "reads" HSBGPG and HSGLTH1 (the same records, used as long reads)
PAF = "\n".join([
    "HSBGPG\t1231\t10\t910\t+\tHSBGPG\t1231\t10\t910\t900\t900\t60",
    "HSBGPG\t1231\t200\t1100\t-\tHSGLTH1\t1020\t50\t950\t850\t900\t60",
    "HSGLTH1\t1020\t0\t300\t+\tHSGLTH1\t1020\t0\t300\t300\t300\t60",   # rejected: span < 400
    "HSGLTH1\t1020\t0\t1\t+\tHSBGPG\t1231\t0\t1\t0\t1\t0",             # last line: never parsed
]) + "\n"


def test_cfg1_plumbing(oracle, tmp_path):
    path = tmp_path / "cfg1.paf"
    path.write_text(PAF)
    paf = overlap.parse_paf(str(path))
    want = oracle.parse_paf(str(path))
    assert paf.rows.tobytes() == want["rows"].tobytes()
    assert paf.n_lines == 4 and len(paf.rows) == 2
    assert paf.read_names == ["HSBGPG", "HSGLTH1"] and paf.anchor_names == ["HSBGPG"]

    t = oracle.overlap(paf.rows)
    assert len(t["edges"]) == 1 and len(t["ems"]) == 1 and len(t["orders"]) == 1
    e, em, o = t["edges"][0], t["ems"][0], t["orders"][0]
    assert (e["v1"], e["v2"]) == (0, 1)
    assert (em["ov_lo"], em["ov_hi"]) == (200, 909)          # [max(10,200), min(909,1099)]
    assert em["flags"] == 2                                  # '+' vs '-' -> direction false; both primary
    assert bool(o["flags"] & 8) and not bool(o["flags"] & 4)

    # sequences: the same file as unitig and as nanopore file, ids through the PAF registries
    sa = json.load(open(os.path.join(GOLD, "sa_test_expected.json")))
    f = S.SeqFile(os.path.join(GOLD, "fasta.fa"))
    rid = np.array([paf.read_names.index(n) if n in paf.read_names else 0xffffffff for n in f.names], dtype=np.uint32)
    aid = np.array([paf.anchor_names.index(n) if n in paf.anchor_names else 0xffffffff for n in f.names], dtype=np.uint32)
    st = S.SeqStore(device=-1)
    st.upload(S.NANOPORE, f, rid, len(paf.read_names))
    st.upload(S.ILLUMINA, f, aid, len(paf.anchor_names))
    stores = ("".join(sa["FastaTest"]["sequences"]).encode(),) * 2
    unitig = sa["FastaTest"]["sequences"][0].encode()
    for row, direction in ((paf.rows[0], True), (paf.rows[1], True), (paf.rows[1], False)):
        pieces, n = st.seg_anchor(row, (int(em["ov_lo"]), int(em["ov_hi"])), direction)
        got = SC.apply_pieces(pieces, stores, P.reverse_complement)
        assert got == oracle.anchor_sequence(row, unitig, (200, 909), direction) and n == len(got) == 711
    # forward read on a '+' match: the anchor is the unitig slice itself (strSlice's inclusive end: 200..910)
    assert SC.apply_pieces(st.seg_anchor(paf.rows[0], (200, 909), True)[0], stores, P.reverse_complement) == unitig[200:911]
    st.close()
    # the FASTQ index path of the same config
    fq = S.SeqFile(os.path.join(GOLD, "fastq.fq"))
    assert [fq.sequence(i).decode() for i in range(len(fq))] == sa["FastQTest"]["nanopore"]


import pytest  # noqa: E402


@pytest.mark.gpu
def test_cfg1_on_the_gpu(oracle, tmp_path):
    """The GPU-visible half of BASELINE.json configs[0]: the same four PAF lines through the loader, the HBM index build, the
    candidate scan and the chain kernels (msgpu_load_rows .. msgpu_chaining_and_overlaps, then the resident dispatcher and a
    group of one) == the oracle's tables in every field; and the reference's test_data/fasta.fa through the HBM sequence
    store and the gather kernel (both store forms): the anchor segments of the job's one EdgeMatch, byte for byte."""
    import torch
    from helpers import assert_tables_equal
    path = tmp_path / "cfg1.paf"
    path.write_text(PAF)
    paf = overlap.parse_paf(str(path))
    want = oracle.overlap(paf.rows)
    with overlap.OverlapContext(0) as ctx:
        ctx.load_rows(paf.rows)
        ctx.calculate_edges()
        ctx.chaining_and_overlaps()
        assert_tables_equal(ctx.tables(), want, "cfg1")
        rl, fl = ctx.reads()
        assert list(rl) == [1231, 1020] and list(fl) == [0, 1]
        lean, _ = ctx.overlap_batched(paf.rows, 2, resident=True, edgematches=False)
        assert_tables_equal(dict(lean, ems=ctx.tables()["ems"]), want, "cfg1, resident dispatcher")
    with overlap.OverlapGroup([0]) as grp:
        t, info = grp.overlap(paf.rows)
        assert_tables_equal(dict(t, ems=want["ems"]), want, "cfg1, group of one")
    em = want["ems"][0]
    sa = json.load(open(os.path.join(GOLD, "sa_test_expected.json")))
    unitig = sa["FastaTest"]["sequences"][0].encode()
    f = S.SeqFile(os.path.join(GOLD, "fasta.fa"))
    rid = np.array([paf.read_names.index(n) if n in paf.read_names else 0xffffffff for n in f.names], dtype=np.uint32)
    aid = np.array([paf.anchor_names.index(n) if n in paf.anchor_names else 0xffffffff for n in f.names], dtype=np.uint32)
    for packed in (False, True):
        st = S.SeqStore(device=0)
        st.upload(S.NANOPORE, f, rid, len(paf.read_names))
        st.upload(S.ILLUMINA, f, aid, len(paf.anchor_names))
        if packed:
            st.pack()
        for row, direction in ((paf.rows[0], True), (paf.rows[1], True), (paf.rows[1], False)):
            pieces, n = st.seg_anchor(row, (int(em["ov_lo"]), int(em["ov_hi"])), direction)
            plan = st.plan(pieces)
            out = torch.full((n + 64,), 0x2e, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()  # (the store's stream is its own: include/msgpu.h, STREAM CONTRACT rule 3)
            st.run(plan, out.data_ptr(), n)
            st.synchronize()
            host = out.cpu().numpy()
            assert host[:n].tobytes() == oracle.anchor_sequence(row, unitig, (200, 909), direction) and (host[n:] == 0x2e).all()
        st.close()
