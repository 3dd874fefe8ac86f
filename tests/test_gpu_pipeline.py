"""The whole flow on the GPU box: PAF text + FASTA/FASTQ files -> temp_1.target.fa / query.fa / align.paf
(muchsalsa_amd.pipeline = src/main.cpp:130-322), against the same flow made of oracles only (C overlap oracle, C
findContractionEdges, Python graph stage, Python assemblePath) and against the genome the reads were cut from."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from graphcases import varlen_rows

pytestmark = pytest.mark.gpu
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def make_dataset(d, seed, jitter, fastq):
    lay = {}
    rows = varlen_rows(400, 0, 250_000, seed, tiled=True, layout=lay, jitter=jitter)
    genome = np.random.default_rng(99 + seed).choice(np.frombuffer(b"ACGT", dtype=np.uint8), 250_000).tobytes()
    nano, illu = {}, {}
    for i in range(len(lay["r_start"])):
        s = genome[int(lay["r_start"][i]): int(lay["r_start"][i]) + int(lay["r_len"][i])]
        nano[i] = s if lay["r_fwd"][i] else s.translate(_COMP)[::-1]
    for j in range(len(lay["a_start"])):
        illu[j] = genome[int(lay["a_start"][j]): int(lay["a_start"][j]) + int(lay["a_len"][j])]
    with open(d / "contigs.paf", "w") as f:  # one line per row, in line order, + the line the reference never parses
        for r in rows:
            a, rd = int(r["anchor_id"]), int(r["read_id"])
            f.write("u%d\t%d\t%d\t%d\t%s\tr%d\t%d\t%d\t%d\t%d\t%d\t60\n" % (
                a, len(illu[a]), r["i_lo"], int(r["i_hi"]) + 1, "+" if int(r["flags"]) & 1 else "-", rd, r["read_len"],
                r["n_lo"], int(r["n_hi"]) + 1, r["score"], int(r["i_hi"]) + 1 - int(r["i_lo"])))
        f.write("u0\t1\t0\t1\t+\tr0\t1\t0\t1\t0\t1\t0\n")
    with open(d / "unitigs.fa", "wb") as f:
        for j in sorted(illu, reverse=True):  # file order is unrelated to Registry order
            f.write(b">u%d some description\n" % j)
            for k in range(0, len(illu[j]), 70):
                f.write(illu[j][k:k + 70] + b"\n")
    name = "nanopore.fq" if fastq else "nanopore.fa"
    with open(d / name, "wb") as f:
        for i in sorted(nano):
            if fastq:
                f.write(b"@r%d\n" % i + nano[i] + b"\n+\n" + b"I" * len(nano[i]) + b"\n")
            else:
                f.write(b">r%d\n" % i + nano[i] + b"\n")
    return rows, lay, genome, nano, illu, name


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
        bound = 8 * longest["n_anchors"]
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
