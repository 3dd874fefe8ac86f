"""The two independent CPU restatements (C: oracle/ms_oracle.c, Python: oracle/ms_oracle_py.py) must agree bit for bit.

Neither is pinned by a reference fixture (the reference's tests hold none for this path); agreement of two
restatements written separately from the reference text is the strongest check available besides the recorded
aggregates of test_oracle_survey_counts.py.
"""
import numpy as np
import pytest

import ms_oracle_py as P
from muchsalsa_amd import synth


def _compare(rows, oracle):
    c = oracle.overlap(rows)
    py_edges, counters = P.overlap(rows)
    assert len(py_edges) == len(c["edges"])
    assert counters.get("compat", 0) == c["compat_checks"]
    ids = c["ids"]
    for pe, ce in zip(py_edges, c["edges"]):
        assert (pe["v1"], pe["v2"], int(pe["shadow"])) == (int(ce["v1"]), int(ce["v2"]), int(ce["shadow"]))
        cems = c["ems"][int(ce["em_off"]):int(ce["em_off"]) + int(ce["em_cnt"])]
        assert len(pe["ems"]) == len(cems)
        for a, b in zip(pe["ems"], cems):
            assert (a["anchor_id"], a["ov_lo"], a["ov_hi"], a["flags"], a["line"]) == (
                int(b["anchor_id"]), int(b["ov_lo"]), int(b["ov_hi"]), int(b["flags"]), int(b["line"]))
            assert float(a["score"]).hex() == float(b["score"]).hex()
        cords = c["orders"][int(ce["order_off"]):int(ce["order_off"]) + int(ce["order_cnt"])]
        assert len(pe["orders"]) == len(cords)
        for a, b in zip(pe["orders"], cords):
            fl = int(b["flags"])
            assert (a["start"], a["end"], a["base"]) == (int(b["start"]), int(b["end"]), int(b["base"]))
            assert (a["contained"], a["direction"], a["primary"]) == (bool(fl & 2), bool(fl & 4), bool(fl & 8))
            assert a["score"] == int(b["score"])
            assert float(a["left"]).hex() == float(b["left_offset"]).hex()
            assert float(a["right"]).hex() == float(b["right_offset"]).hex()
            assert a["ids"] == [int(x) for x in ids[int(b["ids_off"]):int(b["ids_off"]) + int(b["ids_cnt"])]]


@pytest.mark.parametrize("shape", [(60, 3000, 150, 1), (120, 4000, 400, 2), (80, 8000, 500, 3), (200, 2500, 500, 4)])
def test_c_and_python_restatements_agree(oracle, shape):
    _compare(synth.synth_rows(*shape), oracle)


def test_agree_with_duplicates_and_shuffled_rows(oracle):
    rows = synth.synth_rows(100, 4000, 300, 9)
    rng = np.random.default_rng(5)
    # duplicate some (read, anchor) rows with new (later) line numbers and different coordinates
    dup = rows[rng.choice(len(rows), 40, replace=False)].copy()
    dup["line"] = rows["line"].max() + 1 + np.arange(len(dup))
    dup["n_lo"] += 7
    allrows = np.concatenate([rows, dup])
    rng.shuffle(allrows)
    _compare(allrows, oracle)
    # the lowest line wins: results equal those of the un-duplicated table
    a, b = oracle.overlap(allrows), oracle.overlap(rows)
    for k in ("edges", "ems", "orders", "ids"):
        assert a[k].tobytes() == b[k].tobytes()


def test_parse_agrees(oracle, tmp_path):
    tab = synth.paf_table(80, 3000, 200, 12)
    lines = synth.paf_lines(tab)
    path = tmp_path / "in.paf"
    path.write_text("\n".join(lines) + "\n")
    c = oracle.parse_paf(str(path))
    py_rows, rn, an = P.parse_paf_text(path.read_text())
    assert len(py_rows) == len(c["rows"]) and rn == c["read_names"] and an == c["anchor_names"]
    for a, b in zip(py_rows, c["rows"]):
        assert all(a[k] == int(b[k]) for k in a)
    # and both agree with the vectorised generator's own A1 restatement
    rows, rn2, an2 = synth.accepted_rows(tab)
    assert rows.tobytes() == c["rows"].tobytes() and rn2 == rn and an2 == an
