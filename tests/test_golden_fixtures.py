"""The oracle must keep reproducing the committed golden tables (tests/golden/*.npz, tools/make_golden.py)."""
import glob
import os

import numpy as np

from helpers import assert_tables_equal


def test_oracle_reproduces_golden_tables(oracle):
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert len(files) >= 2
    for f in files:
        z = np.load(f)
        got = oracle.overlap(z["rows"])
        assert_tables_equal(got, {k: z[k] for k in ("edges", "ems", "orders", "ids")}, os.path.basename(f))
