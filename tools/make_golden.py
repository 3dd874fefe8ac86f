#!/usr/bin/env python3
"""Golden fixtures for tests/golden/: inputs + the CPU oracle's tables (oracle/ms_oracle.c).

The reference cannot be built in this image (its un-vendored GSL dependency is absent), so these vectors come from
the oracle, whose two independent restatements agree and which reproduces the aggregate counts the survey recorded
from the real reference (tests/test_oracle_survey_counts.py).  They pin the oracle against regressions and let the
GPU tests run against committed data.  Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import ms_oracle_ctypes as oracle  # noqa: E402
from muchsalsa_amd import synth  # noqa: E402

CASES = {
    # name: (n_reads, read_len, n_anchors, seed)
    "synth_150x4k_s21": (150, 4000, 500, 21),
    "synth_250x8k_s22": (250, 8000, 1600, 22),   # has edges with > 64 EdgeMatches (big-edge kernel)
}


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, shape in CASES.items():
        rows = synth.synth_rows(*shape)
        t = oracle.overlap(rows)
        np.savez_compressed(os.path.join(out, name + ".npz"), rows=rows, edges=t["edges"], ems=t["ems"],
                            orders=t["orders"], ids=t["ids"])
        print(name, len(rows), "rows", len(t["edges"]), "edges", len(t["ems"]), "ems", len(t["orders"]), "orders",
              "max em_cnt", int(t["edges"]["em_cnt"].max()) if len(t["edges"]) else 0)


if __name__ == "__main__":
    main()
