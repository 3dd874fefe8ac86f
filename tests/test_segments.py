"""Segment builders of assemblePath (ap.cpp:352-579): the two restatements agree, and libmsgpu's piece composers
(host code; layout-only context, no GPU) describe exactly the same strings."""
import numpy as np
import pytest

import ms_oracle_py as P
import segcases as SC
from muchsalsa_amd import sequences as S


def _store(tmp_path, reads, unis):
    def write(path, seqs, prefix):
        with open(path, "w") as f:
            for i, s in enumerate(seqs):
                f.write(">%s%d\n%s\n" % (prefix, i, s.decode()))
    write(tmp_path / "r.fa", reads, "r")
    write(tmp_path / "u.fa", unis, "u")
    st = S.SeqStore(device=-1)  # layout-only: no device needed
    fr, fu = S.SeqFile(str(tmp_path / "r.fa")), S.SeqFile(str(tmp_path / "u.fa"))
    st.upload(S.NANOPORE, fr)
    st.upload(S.ILLUMINA, fu)
    # (what a device store would hold: the loader's buffer as it is -- a file parsed in stretches has unused bytes between them)
    return st, (fr.buffer(), fu.buffer())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_restatements_and_composers_agree(oracle, tmp_path, seed):
    rng, reads, unis = SC.make_world(seed)
    st, stores = _store(tmp_path, reads, unis)
    for _ in range(300):
        m, ov = SC.random_match(rng, reads, unis)
        nano, illu = reads[int(m["read_id"])], unis[int(m["anchor_id"])]
        for direction in (True, False):
            for name, c_fn, p_fn, comp in (
                    ("anchor", lambda: oracle.anchor_sequence(m, illu, ov, direction),
                     lambda: P.get_anchor_sequence(m, illu, ov, direction),
                     lambda: st.seg_anchor(m, ov, direction)),
                    ("left", lambda: oracle.left_of_anchor(m, nano, illu, len(nano), ov, direction),
                     lambda: P.get_sequence_left_of_anchor(m, nano, illu, len(nano), ov, direction),
                     lambda: st.seg_left_of_anchor(m, len(nano), ov, direction)),
                    ("right", lambda: oracle.right_of_anchor(m, nano, illu, len(nano), ov, direction),
                     lambda: P.get_sequence_right_of_anchor(m, nano, illu, len(nano), ov, direction),
                     lambda: st.seg_right_of_anchor(m, len(nano), ov, direction))):
                want = c_fn()
                assert want == p_fn(), name
                pieces, length = comp()
                assert length == len(want), name
                assert SC.apply_pieces(pieces, stores, P.reverse_complement) == want, name
        # between two anchors of one read
        m2, ov2 = SC.random_match(rng, reads, unis, read_id=int(m["read_id"]))
        for direction in (True, False):
            for ml, ovl, mr, ovr in ((m, ov, m2, ov2), (m2, ov2, m, ov)):
                il, ir = unis[int(ml["anchor_id"])], unis[int(mr["anchor_id"])]
                dist, seq = oracle.between_anchors(ml, mr, nano, il, ir, ovl, ovr, direction)
                assert (dist, seq) == P.get_sequence_between_anchors(ml, mr, nano, il, ir, ovl, ovr, direction)
                pieces, d2, has = st.seg_between_anchors(ml, mr, ovl, ovr, direction)
                assert d2 == dist and has == (seq is not None)
                if seq is not None:
                    assert SC.apply_pieces(pieces, stores, P.reverse_complement) == seq
    st.close()


def test_between_anchors_hits_every_branch(oracle):
    """The correction branches of getSequenceBetweenAnchors (overlapping nanopore ranges, :459-497 / :521-559)."""
    rng, reads, unis = SC.make_world(9, n_reads=4, n_unitigs=6)
    seen = set()
    for _ in range(4000):
        m, ov = SC.random_match(rng, reads, unis, read_id=0)
        m2, ov2 = SC.random_match(rng, reads, unis, read_id=0)
        # force nearby anchors so that the raw ranges overlap sometimes
        m2["n_lo"] = max(0, int(m["n_hi"]) + int(rng.integers(-200, 200)))
        m2["n_hi"] = min(len(reads[0]) - 1, int(m2["n_lo"]) + int(m2["i_hi"]) - int(m2["i_lo"]))
        for direction in (True, False):
            ml, mr = (m, m2)
            il, ir = unis[int(ml["anchor_id"])], unis[int(mr["anchor_id"])]
            dist, seq = oracle.between_anchors(ml, mr, reads[0], il, ir, ov, ov2, direction)
            assert (dist, seq) == P.get_sequence_between_anchors(ml, mr, reads[0], il, ir, ov, ov2, direction)
            err = (int(ml["n_hi"]) - int(mr["n_lo"])) if direction else (int(mr["n_hi"]) - int(ml["n_lo"]))
            seen.add((direction, err > 0, seq is None))
    assert {(True, True, True), (True, True, False), (True, False, False)} <= seen
    assert any(k[0] is False and k[1] for k in seen)


def py_update_consensus_base(old, old_b, new, new_b):
    """updateConsensusBase, ap.cpp:205-229, on byte strings."""
    if old is None:
        return new, new_b[0], new_b[1]
    if new_b[0] < old_b[0]:
        upd = P.str_slice(new, 0, old_b[0] - new_b[0]) + old
    elif new_b[1] > old_b[1]:
        upd = old + P.str_slice(new, -(new_b[1] - old_b[1]), len(new))
    else:
        upd = old
    return upd, min(old_b[0], new_b[0]), max(old_b[1], new_b[1])


@pytest.mark.parametrize("seed", [4, 5, 6])
def test_update_consensus_base_on_pieces(oracle, tmp_path, seed):
    """Random walks of updateConsensusBase: libmsgpu's piece version against the string version, including new
    sequences that are shorter/longer than the uncovered stretch (the strSlice clipping cases)."""
    rng, reads, unis = SC.make_world(seed)
    st, stores = _store(tmp_path, reads, unis)
    for _ in range(40):
        cb = S.ConsensusBase()
        want, wb = None, (0, 0)
        for _ in range(int(rng.integers(1, 25))):
            m, ov = SC.random_match(rng, reads, unis)
            nano, illu = reads[int(m["read_id"])], unis[int(m["anchor_id"])]
            d = bool(rng.integers(0, 2))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                seg, s = st.seg_anchor(m, ov, d)[0], oracle.anchor_sequence(m, illu, ov, d)
            elif kind == 1:
                seg, s = st.seg_left_of_anchor(m, len(nano), ov, d)[0], oracle.left_of_anchor(m, nano, illu, len(nano), ov, d)
            else:
                seg, s = st.seg_right_of_anchor(m, len(nano), ov, d)[0], oracle.right_of_anchor(m, nano, illu, len(nano), ov, d)
            # borders: sometimes consistent with the length, sometimes not (then strSlice clips)
            lo = int(rng.integers(-3000, 3000))
            hi = lo + len(s) - 1 + int(rng.choice([0, 0, 0, -7, 11, 400]))
            cb.update(seg, lo, hi)
            want, a, b = py_update_consensus_base(want, wb, s, (lo, hi))
            wb = (a, b)
            blo, bhi, length = cb.borders
            assert (blo, bhi) == wb and length == len(want)
            assert SC.apply_pieces(cb.pieces(), stores, P.reverse_complement) == want
        cb.close()
    st.close()
