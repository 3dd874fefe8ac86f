"""The whole flow on the GPU box: PAF text + FASTA/FASTQ files -> temp_1.target.fa / query.fa / align.paf
(muchsalsa_amd.pipeline = src/main.cpp:130-322), against the same flow made of oracles only (C overlap oracle, C
findContractionEdges, Python graph stage, Python assemblePath) and against the genome the reads were cut from."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from graphcases import make_dataset
from helpers import A9_TOLERANCE_EDITS_PER_ANCHOR_VS_GENOME

pytestmark = pytest.mark.gpu
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def oracle_flow(oracle, rows, nano, illu):
    from oracle import ms_graph_py as G
    from oracle.ms_assemble_py import assemble_path
    t = oracle.overlap(rows)
    co = oracle.find_contraction_edges(t, len(t["read_len"]))
    vm = {(int(r["read_id"]), int(r["anchor_id"])): r for r in rows}
    g, el, oo = G.build_graph(t, t["read_len"], t["read_first_line"])
    contain = G.clean_up(g, el, oo, co, lambda r, a: (r, a) in vm)
    edges, ems = t["edges"], t["ems"]
    eidx = {(int(e["v1"]), int(e["v2"])): i for i, e in enumerate(edges)}

    def em_of(a, b):
        e = edges[eidx[(a, b)]]
        return {int(m["anchor_id"]): (int(m["ov_lo"]), int(m["ov_hi"]))
                for m in ems[int(e["em_off"]): int(e["em_off"]) + int(e["em_cnt"])]}
    oc = {v: [dict(nano=c["nano"], dir=c["direction"], matches={a: vm[(c["nano"], a)] for a in c["anchors"]})
              for c in lst] for v, lst in contain.items()}
    out = [assemble_path(p, st, vm, oc, nano, illu, k) for k, (p, st) in enumerate(G.assemble_all(g, em_of))]
    return out


@pytest.mark.parametrize("seed,jitter,fastq", [(2, 0, False), (3, 10, True)])
def test_files_in_files_out(oracle, tmp_path, seed, jitter, fastq):
    from muchsalsa_amd import pipeline
    rows, lay, genome, nano, illu, nano_name = make_dataset(tmp_path, seed, jitter, fastq)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    res = pipeline.run(str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / nano_name),
                       str(out_dir), threads=4)
    want = oracle_flow(oracle, rows, nano, illu)
    assert res["rows"] == len(rows) and res["contigs"] == len(want) and res["paths_skipped"] == 0
    assert (out_dir / "temp_1.target.fa").read_bytes() == b"".join(r["target_fa"] for r in want)
    assert (out_dir / "temp_1.query.fa").read_bytes() == b"".join(r["query_fa"] for r in want)
    assert (out_dir / "temp_1.align.paf").read_bytes() == b"".join(r["paf"] for r in want)
    # and the assembly is right: one contig spanning the genome, off by the joint duplicates only (jitter 0)
    longest = max(want, key=lambda r: len(r["target"]))
    assert len(longest["target"]) > 0.95 * len(genome)
    if jitter == 0:
        lo, hi = int(lay["r_start"].min()), int((lay["r_start"] + lay["r_len"]).max())
        bound = A9_TOLERANCE_EDITS_PER_ANCHOR_VS_GENOME * longest["n_anchors"]
        d = oracle.edit_distance_banded(longest["target"], genome[lo:hi], bound)
        if d > bound:  # the contig may be the other strand
            d = oracle.edit_distance_banded(longest["target"].translate(_COMP)[::-1], genome[lo:hi], bound)
        assert d <= bound, (d, longest["n_anchors"])


def test_command_line(tmp_path):
    make_dataset(tmp_path, 5, 5, False)
    out_dir = tmp_path / "o"
    out_dir.mkdir()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "muchsalsa_amd", str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"),
                        str(tmp_path / "nanopore.fa"), str(out_dir), "2", "300"], cwd=root, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["contigs"] >= 1 and res["target_bases"] > 200_000
    assert (out_dir / "temp_1.target.fa").stat().st_size > res["target_bases"]
