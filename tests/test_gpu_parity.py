"""GPU parity: libmsgpu (through the C-ABI) against the CPU oracle on identical rows.  Bit-exact."""
import numpy as np
import pytest

from helpers import assert_tables_equal

pytestmark = pytest.mark.gpu


def _gpu_tables(rows, **kw):
    from muchsalsa_amd import overlap
    return overlap.build_overlaps(rows, **kw)


@pytest.mark.parametrize("shape", [(300, 3000, 900, 1), (2000, 5000, 10000, 7), (1500, 10000, 8000, 11)])
def test_synthetic_matches_oracle(oracle, shape):
    from muchsalsa_amd import synth
    rows = synth.synth_rows(*shape)
    want = oracle.overlap(rows)
    got = _gpu_tables(rows)
    assert_tables_equal(got, want, "synth%r" % (shape,))


def test_empty_and_tiny(oracle):
    from muchsalsa_amd import synth
    rows = synth.synth_rows(300, 3000, 900, 1)
    for n in (0, 1, 2, 5):
        sub = rows[:n].copy()
        # keep Registry-dense ids
        _, sub["read_id"] = np.unique(sub["read_id"], return_inverse=True)
        _, sub["anchor_id"] = np.unique(sub["anchor_id"], return_inverse=True)
        assert_tables_equal(_gpu_tables(sub), oracle.overlap(sub), "n=%d" % n)
