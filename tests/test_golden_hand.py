"""Hand-derived known-answer vectors for the overlap path.

Expected values are worked out from the reference text (file:line in the comments) with explicit arithmetic here,
independently of either restatement's code.  They check the CPU oracle on CPU; test_gpu_parity.py re-uses CASES to
check the HIP path against the same expectations on the GPU.
"""
import numpy as np
import pytest

from muchsalsa_amd.synth import ROW_DTYPE


def row(anchor, read, read_len, i_lo, i_hi, n_lo, n_hi, score, line, plus):
    prim = (i_hi - i_lo + 1) >= 500 and score >= 500  # BlastFileReader.cpp:121-122
    return (anchor, read, read_len, i_lo, i_hi, n_lo, n_hi, score, line, (1 if plus else 0) | (2 if prim else 0))


def case_single_anchor(plus1=True, i1=(100, 699)):
    """Two reads sharing one anchor."""
    rows = np.array([row(0, 0, 5000, 0, 599, 1000, 1599, 550, 0, True),
                     row(0, 1, 5000, i1[0], i1[1], 200, 799, 560, 1, plus1)], dtype=ROW_DTYPE)
    return rows


def expected_single_anchor(plus1):
    # MatchMap.cpp:188-202: outer = line 1 (read 1), inner = line 0 (read 0)
    ov = (max(100, 0), min(699, 599))
    assert ov == (100, 599) and ov[1] - ov[0] > 100
    cl, ol, il = float(ov[1] - ov[0] + 1), float(699 - 100 + 1), float(599 - 0 + 1)
    score = 560.0 * cl / ol + 550.0 * cl / il
    # getOverhangs (ol.cpp:31-50); rRatio = 600/600 on both reads
    rr = 600.0 / 600.0
    ncl1, ncr1 = (100 - 0) / rr, (599 - 599) / rr          # read 0 is '+'
    L1, R1 = 1000.0 + ncl1, float(5000 - 1599) + ncr1
    ncl2, ncr2 = (100 - 100) / rr, (699 - 599) / rr
    if not plus1:                                           # swap_if(!vertexMatch->direction), ol.cpp:42
        ncl2, ncr2 = ncr2, ncl2
    l2, r2 = 200.0 + ncl2, float(5000 - 799) + ncr2
    if plus1:
        L2, R2 = l2, r2
    else:                                                   # ol.cpp:73-76: second read's sides swap
        L2, R2 = r2, l2
    return dict(score=score, direction=plus1, L1=L1, R1=R1, L2=L2, R2=R2)


def order_from_overhangs(L1, R1, L2, R2):
    """The four cases of ol.cpp:79-97 -> (start_is_v1, contained, left, right)."""
    if L1 <= L2 and R1 <= R2:
        return True, True, L2 - L1, R2 - R1
    if L1 >= L2 and R1 >= R2:
        return False, True, L1 - L2, R1 - R2
    if L1 > L2 and R1 < R2:
        return True, False, L1 - L2, R2 - R1
    if L1 < L2 and R1 > R2:
        return False, False, L2 - L1, R1 - R2
    return None


def check_single_anchor(t, plus1):
    exp = expected_single_anchor(plus1)
    assert len(t["edges"]) == 1 and len(t["ems"]) == 1 and len(t["orders"]) == 1 and list(t["ids"]) == [0]
    e, em, o = t["edges"][0], t["ems"][0], t["orders"][0]
    # v1 = read with the lower first line (MatchMap.cpp:204-213)
    assert (e["v1"], e["v2"], e["em_cnt"], e["order_cnt"]) == (0, 1, 1, 1)
    assert (em["ov_lo"], em["ov_hi"], em["anchor_id"], em["line"]) == (100, 599, 0, 1)
    assert em["flags"] == (1 if plus1 else 0) | 2
    assert float(em["score"]) == exp["score"]
    # single anchor: hasPrimary = EdgeMatch.isPrimary (mpp.cpp:217-220); anchored at both read ends (both reads have
    # only this anchor) -> stays primary (:272-296) -> edge is not a shadow (main.cpp:393-394)
    assert e["shadow"] == 0
    start_v1, contained, left, right = order_from_overhangs(exp["L1"], exp["R1"], exp["L2"], exp["R2"])
    fl = int(o["flags"])
    assert bool(fl & 1) == start_v1 and bool(fl & 2) == contained and bool(fl & 4) == plus1 and bool(fl & 8)
    assert float(o["left_offset"]) == left and float(o["right_offset"]) == right
    assert int(o["score"]) == int(exp["score"])  # truncation to std::size_t (mpp.cpp:34,221)
    assert (o["start"], o["end"], o["base"]) == ((0, 1, 0) if start_v1 else (1, 0, 0))
    assert (o["ids_off"], o["ids_cnt"]) == (0, 1)


def test_expected_numbers_by_hand():
    # the worked example in numbers: score = 560*500/600 + 550*500/600, overhangs 1100/3401 vs 200/4301
    exp = expected_single_anchor(True)
    assert abs(exp["score"] - 925.0) < 1e-9
    assert (exp["L1"], exp["R1"], exp["L2"], exp["R2"]) == (1100.0, 3401.0, 200.0, 4301.0)
    assert order_from_overhangs(1100.0, 3401.0, 200.0, 4301.0) == (True, False, 900.0, 900.0)
    expm = expected_single_anchor(False)
    assert (expm["L2"], expm["R2"]) == (4201.0, 300.0)  # read 1 flipped: (len-n_hi)+0 = 4201 on the left, 200+100 right
    assert order_from_overhangs(1100.0, 3401.0, 4201.0, 300.0) == (False, False, 3101.0, 3101.0)


@pytest.mark.parametrize("plus1", [True, False])
def test_single_anchor_oracle(oracle, plus1):
    check_single_anchor(oracle.overlap(case_single_anchor(plus1)), plus1)


def case_overlap_threshold(delta):
    """ov.hi - ov.lo == delta: an EdgeMatch exists iff delta > 100 (strict, MatchMap.cpp:192)."""
    return np.array([row(0, 0, 5000, 0, 599, 1000, 1599, 550, 0, True),
                     row(0, 1, 5000, 599 - delta, 1200, 200, 801 + delta, 560, 1, True)], dtype=ROW_DTYPE)


@pytest.mark.parametrize("delta,edges", [(99, 0), (100, 0), (101, 1), (400, 1)])
def test_overlap_threshold_oracle(oracle, delta, edges):
    t = oracle.overlap(case_overlap_threshold(delta))
    assert len(t["edges"]) == edges and len(t["ems"]) == edges


def case_two_anchor_chain(gap2):
    """Two reads, two shared anchors 2000 bp apart on read 0 and `gap2` apart on read 1 (all '+', rRatio 1)."""
    return np.array([
        row(0, 0, 9000, 0, 599, 1000, 1599, 580, 0, True),
        row(0, 1, 9000, 0, 599, 3000, 3599, 570, 1, True),
        row(1, 0, 9000, 0, 599, 3600, 4199, 560, 2, True),
        row(1, 1, 9000, 0, 599, 3600 + gap2, 4199 + gap2, 550, 3, True),
    ], dtype=ROW_DTYPE)


@pytest.mark.parametrize("gap2,chained", [(2000, True), (2300, True), (2301, True), (2353, True), (2354, False), (3000, False)])
def test_two_anchor_chain_oracle(oracle, gap2, chained):
    """checkCompatibility (mpp.cpp:133-139): orientation 1 on both reads, diff1 = 3600-1599+1 = 2002,
    diff2 = 2002 + (gap2 - 2000).  Compatible iff |d1-d2| <= 300 or |d1-d2|*100/max <= 15:
    301*100/2303 = 13.07 and 353*100/2355 = 14.99 still chain, 354*100/2356 = 15.03 does not."""
    d1, d2 = 3600.0 - 1599.0 + 1, (3600.0 + gap2) - 3599.0 + 1
    df = max(d1, d2) - min(d1, d2)
    assert chained == (df <= 300.0 or df * 100 / max(d1, d2) <= 15)
    t = oracle.overlap(case_two_anchor_chain(gap2))
    assert len(t["edges"]) == 1 and len(t["ems"]) == 2
    s0, s1 = 570.0 * 600.0 / 600.0 + 580.0 * 600.0 / 600.0, 550.0 + 560.0
    assert [float(x) for x in t["ems"]["score"]] == [s0, s1]
    if chained:
        # one path [u0, u1], score truncated; len 2 -> not "> 2", primary through EdgeMatch.isPrimary
        assert len(t["orders"]) == 1 and list(t["ids"]) == [0, 1] and int(t["orders"][0]["score"]) == int(s0 + s1)
        assert t["edges"][0]["shadow"] == 0
    else:
        # two disjoint single-anchor paths: best = u0 (1150 > 1110), alternative u1 (1110 > 0.75*1150);
        # combined size 2 -> shadow (main.cpp:389-391)
        assert len(t["orders"]) == 2 and list(t["ids"]) == [0, 1]
        assert [int(s) for s in t["orders"]["score"]] == [int(s0), int(s1)]
        assert t["edges"][0]["shadow"] == 1


CASES = {
    "single_plus": case_single_anchor(True), "single_minus": case_single_anchor(False),
    "thr100": case_overlap_threshold(100), "thr101": case_overlap_threshold(101),
    "chain2000": case_two_anchor_chain(2000), "chain2300": case_two_anchor_chain(2300),
    "chain2301": case_two_anchor_chain(2301), "chain2353": case_two_anchor_chain(2353),
    "chain2354": case_two_anchor_chain(2354), "chain3000": case_two_anchor_chain(3000),
}
