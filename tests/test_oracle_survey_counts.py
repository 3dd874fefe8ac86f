"""Cross-check of the CPU oracle against aggregates the survey recorded from the REAL reference.

SURVEY.md section 8 lists, for the cfg2-shaped input of its generator (seed 42: 515,324 PAF data rows), the counts the
reference itself produced at one thread.  tools/survey_gen.py re-creates that input (same CPython MT19937 draw order),
so the oracle can be checked against every recorded aggregate.  Counts, not tables: a cross-check, not a pin.
"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# SURVEY.md section 8 (config 2, measured with the reference at 1 thread)
RECORDED = dict(paf_rows=515_324, R=511_510, A=49_996, P_eval=2_623_065, P_emit=2_610_522, E=97_776, C=44_522_507,
                O=112_250, shadow_edges=95_979, sum_ids=1_035_911, max_ids_per_order=30)


def test_oracle_reproduces_survey_recorded_reference_counts(oracle, tmp_path):
    spec = importlib.util.spec_from_file_location("survey_gen", os.path.join(ROOT, "tools", "survey_gen.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    lines = gen.generate(10_000, 5_000, 50_000, 5_000_000, 42)
    assert len(lines) - 1 == RECORDED["paf_rows"]
    path = tmp_path / "cfg2.paf"
    path.write_text("\n".join(lines) + "\n")
    parsed = oracle.parse_paf(str(path))
    rows = parsed["rows"]
    t = oracle.overlap(rows)
    got = dict(paf_rows=parsed["n_lines"] - 1, R=len(rows), A=t["n_anchors"], P_eval=t["p_eval"], P_emit=len(t["ems"]),
               E=len(t["edges"]), C=t["compat_checks"], O=len(t["orders"]), shadow_edges=t["shadow_edges"],
               sum_ids=len(t["ids"]), max_ids_per_order=int(t["orders"]["ids_cnt"].max()))
    assert got == RECORDED
