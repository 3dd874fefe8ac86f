#!/usr/bin/env python3
"""Differential run of the WHOLE flow on the GPU box: random small data sets (tiled anchors, random read lengths and
strands, FASTA or FASTQ reads) written as files, through muchsalsa_amd.pipeline.run with random numbers of parser chunks /
graph-stage threads (the threaded host paths at small sizes), against the flow made of oracles only (C overlap oracle + C
findContractionEdges + Python graph stage + Python assemblePath): the three output files must be byte-identical.
    python tools/fuzz_pipeline.py [n_cases] [first_seed]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import ms_oracle_ctypes as O  # noqa: E402
from graphcases import make_dataset  # noqa: E402
from muchsalsa_amd import pipeline  # noqa: E402
from test_gpu_pipeline import oracle_flow  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    O.build()
    for case in range(n_cases):
        seed = seed0 + case
        rng = np.random.default_rng(seed)
        n_reads, genome_len = int(rng.integers(60, 500)), int(rng.integers(40_000, 300_000))
        jitter, fastq = int(rng.integers(0, 12)), bool(rng.integers(0, 2))
        # the host stages' threaded paths at these sizes (they read the environment at every call)
        os.environ["MSGPU_PARSE_THREADS"] = str(int(rng.integers(1, 9)))
        os.environ["MSGPU_SEQ_THREADS"] = str(int(rng.integers(1, 7)))
        os.environ["MSGPU_GRAPH_THREADS"] = str(int(rng.integers(1, 9)))
        os.environ["MSGPU_GRAPH_PAR_MIN"] = str(int(rng.choice([32, 64, 1 << 16])))
        with tempfile.TemporaryDirectory() as d:
            d = pathlib.Path(d)
            rows, lay, genome, nano, illu, name = make_dataset(d, seed, jitter, fastq, n_reads, genome_len)
            (d / "out").mkdir()
            res = pipeline.run(str(d / "contigs.paf"), str(d / "unitigs.fa"), str(d / name), str(d / "out"),
                               threads=int(rng.integers(1, 6)))
            want = oracle_flow(O, rows, nano, illu)
            for key, fn in (("target_fa", "temp_1.target.fa"), ("query_fa", "temp_1.query.fa"), ("paf", "temp_1.align.paf")):
                got = (d / "out" / fn).read_bytes()
                assert got == b"".join(r[key] for r in want), "case %d (seed %d): %s differs" % (case, seed, fn)
            assert res["contigs"] == len(want)
        print("case %3d ok: %4d reads, genome %6d, jitter %2d, %s, %3d rows -> %2d contigs, %7d bases" % (
            case, n_reads, genome_len, jitter, "fq" if fastq else "fa", len(rows), res["contigs"], res["target_bases"]), flush=True)


if __name__ == "__main__":
    main()
