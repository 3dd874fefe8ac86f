"""tests/golden/pipeline_flow.json: SHA-256 of target.fa / query.fa / align.paf of the whole flow on fixed generated
data sets (tools/make_pipeline_golden.py).  CPU: the oracle-only flow still produces them; GPU: so does the product."""
import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "pipeline_flow.json")))


@pytest.mark.parametrize("want", GOLDEN, ids=lambda w: "seed%d" % w["seed"])
def test_oracle_flow_reproduces_golden(oracle, want):
    from make_pipeline_golden import flow_digest
    got = flow_digest({k: want[k] for k in ("seed", "jitter", "n_reads", "genome_len")})
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("want", GOLDEN, ids=lambda w: "seed%d" % w["seed"])
def test_product_reproduces_golden(tmp_path, want):
    from graphcases import make_dataset
    from muchsalsa_amd import pipeline
    make_dataset(tmp_path, want["seed"], want["jitter"], want["seed"] % 2 == 1, want["n_reads"], want["genome_len"])
    name = "nanopore.fq" if want["seed"] % 2 == 1 else "nanopore.fa"
    out = tmp_path / "out"
    out.mkdir()
    res = pipeline.run(str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / name), str(out), threads=3)
    assert (res["rows"], res["contigs"], res["target_bases"], res["queries"]) == (
        want["rows"], want["contigs"], want["target_bases"], want["queries"])
    got = [hashlib.sha256((out / n).read_bytes()).hexdigest()
           for n in ("temp_1.target.fa", "temp_1.query.fa", "temp_1.align.paf")]
    assert got == want["sha256"]
