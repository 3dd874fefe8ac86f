"""Shared helpers of the parity tests."""
import numpy as np

TABLES = ("edges", "ems", "orders", "ids")


def assert_tables_equal(got, want, what=""):
    """Bit-exact comparison of result tables (ints exact, doubles bitwise)."""
    for name in TABLES:
        g, w = got[name], want[name]
        assert len(g) == len(w), "%s %s: %d rows, oracle has %d" % (what, name, len(g), len(w))
        if g.dtype.names:
            for f in g.dtype.names:
                if f == "pad":
                    continue
                gv, wv = g[f], w[f]
                if gv.dtype.kind == "f":
                    gv, wv = gv.view("<u8"), wv.view("<u8")
                bad = np.nonzero(gv != wv)[0]
                assert len(bad) == 0, "%s %s.%s differs at %d rows, first %d: got %r want %r" % (
                    what, name, f, len(bad), bad[0], g[bad[0]], w[bad[0]])
        else:
            bad = np.nonzero(g != w)[0]
            assert len(bad) == 0, "%s %s differs at %d entries, first %d" % (what, name, len(bad), bad[0])


# north_star: "consensus sequences within a stated edit-distance tolerance" ("canonical order").
#  * against the restatement of ap.cpp (oracle/ms_assemble_py.py) on the same input: edit distance 0 -- the three output
#    texts are compared byte for byte;
#  * against the genome the reads were cut from, with exact PAF coordinates: at most this many edits per placed anchor
#    (the reference's inclusive slices duplicate a few bases at every anchor joint, SequenceUtils.cpp:27-38).
A9_TOLERANCE_VS_RESTATEMENT = 0
A9_TOLERANCE_EDITS_PER_ANCHOR_VS_GENOME = 8
