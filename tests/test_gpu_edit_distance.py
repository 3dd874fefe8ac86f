"""Banded edit-distance kernel (A10: the meter for the consensus tolerance) against the full-DP oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def mutate(rng, s, n_edits):
    s = bytearray(s)
    for _ in range(n_edits):
        op = int(rng.integers(0, 3))
        pos = int(rng.integers(0, len(s) + 1))
        if op == 0 and len(s):  # substitution
            pos = min(pos, len(s) - 1)
            s[pos] = b"ACGT"[(b"ACGT".index(s[pos]) + 1 + int(rng.integers(0, 3))) % 4] if s[pos] in b"ACGT" else 65
        elif op == 1:  # insertion
            s.insert(pos, b"ACGT"[int(rng.integers(0, 4))])
        elif len(s):  # deletion
            del s[min(pos, len(s) - 1)]
    return bytes(s)


def _device_pairs(pairs_ab):
    import torch
    from muchsalsa_amd._lib import ALIGN_PAIR_DTYPE
    a = b"".join(p[0] for p in pairs_ab)
    b = b"".join(p[1] for p in pairs_ab)
    desc = np.zeros(len(pairs_ab), dtype=ALIGN_PAIR_DTYPE)
    ao = bo = 0
    for i, (x, y) in enumerate(pairs_ab):
        desc[i] = (ao, bo, len(x), len(y))
        ao += len(x)
        bo += len(y)
    da = torch.frombuffer(bytearray(a + b"\0"), dtype=torch.uint8).cuda()
    db = torch.frombuffer(bytearray(b + b"\0"), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    return da, db, desc


@pytest.mark.parametrize("kernel", ["furthest-reaching", "anti-diagonal DP"])
@pytest.mark.parametrize("band", [127, 64, 40, 8, 0])
def test_random_pairs_match_full_dp(oracle, band, kernel, monkeypatch):
    """Both kernels behind msgpu_edit_distance (k_edit_distance: furthest-reaching points, the default; k_edit_distance_dp:
    the banded anti-diagonal DP, MSGPU_ED_DP=1) give the full DP's min(d, band + 1)."""
    from muchsalsa_amd import sequences as S
    monkeypatch.setenv("MSGPU_ED_DP", "1" if kernel == "anti-diagonal DP" else "0")
    rng = np.random.default_rng(100 + band)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    pairs = []
    for _ in range(150):
        n = int(rng.integers(0, 1800))
        a = bytes(rng.choice(alpha, n))
        b = mutate(rng, a, int(rng.integers(0, 2 * band + 12)))
        pairs.append((a, b))
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"ACGT", b"TGCA"), (b"A" * 300, b"A" * 290)]
    # unrelated sequences: far beyond the band
    pairs += [(bytes(rng.choice(alpha, 700)), bytes(rng.choice(alpha, 650))) for _ in range(5)]
    da, db, desc = _device_pairs(pairs)
    with S.SeqStore(0) as st:
        got = st.edit_distance(da.data_ptr(), db.data_ptr(), desc, band)
    want = [oracle.edit_distance(a, b, band) for a, b in pairs]
    assert list(got) == want


def test_long_pairs_and_properties(oracle):
    """Sequences longer than the LDS staging area take the global-memory path; distance properties hold."""
    from muchsalsa_amd import sequences as S
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = bytes(rng.choice(alpha, 30000))
    b = mutate(rng, a, 60)
    c = mutate(rng, b, 50)
    short = bytes(rng.choice(alpha, 9000))
    short2 = mutate(rng, short, 90)
    pairs = [(a, b), (b, a), (a, a), (b, c), (a, c), (short, short2), (short2, short)]
    da, db, desc = _device_pairs(pairs)
    with S.SeqStore(0) as st:
        d = [int(x) for x in st.edit_distance(da.data_ptr(), db.data_ptr(), desc, 127)]
    assert d[0] == d[1] and d[2] == 0 and d[5] == d[6]                     # symmetry, identity
    assert d[0] <= 60 and d[3] <= 50 and d[5] <= 90                        # at most the edits applied
    assert d[4] <= d[0] + d[3]                                              # triangle inequality
    assert d[5] == oracle.edit_distance(short, short2, 127)                # 9 kb pair against the full DP


def test_slides_end_exactly_where_the_sequences_differ(oracle, monkeypatch):
    """The furthest-reaching kernel compares 8 bytes per lane and 512 per wavefront round: single edits at every kind of
    position relative to those steps (first and last byte, around multiples of 8, 16 and 512), sequences that end inside a
    step, a pair that ends with the device buffer, and near-identical pairs of a few 10 kb -- against the full DP, and the
    banded anti-diagonal kernel must give the same numbers."""
    from muchsalsa_amd import sequences as S
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = bytes(rng.choice(alpha, 2100))
    pairs = []
    for pos in (0, 1, 7, 8, 9, 15, 16, 17, 23, 24, 511, 512, 513, 519, 520, 527, 528, 529, 1023, 1024, 1031, 1040, 2098, 2099):
        sub = bytearray(base)
        sub[pos] = b"ACGT"[(b"ACGT".index(sub[pos]) + 1) % 4]
        pairs.append((base, bytes(sub)))                     # one substitution
        pairs.append((base, base[:pos] + base[pos + 1:]))    # one deletion
        pairs.append((base[:pos] + b"G" + base[pos:], base))  # one insertion on the other side
    for n in (1, 7, 8, 9, 15, 16, 17, 31, 511, 512, 513, 527, 528, 1031):
        pairs.append((base[:n], base[:n]))
        pairs.append((base[:n], base[:n] + b"A"))
        pairs.append((base[:n] + b"C", base[:n] + b"T"))
    long_a = bytes(rng.choice(alpha, 40000))
    pairs += [(long_a, mutate(rng, long_a, k)) for k in (0, 1, 5, 30)]
    pairs.append((base, base))  # (last: its final bytes are the last of the device buffers but one)
    da, db, desc = _device_pairs(pairs)
    want = [oracle.edit_distance_banded(a, b, 64) if len(a) > 5000 else oracle.edit_distance(a, b, 64) for a, b in pairs]
    with S.SeqStore(0) as st:
        got = st.edit_distance(da.data_ptr(), db.data_ptr(), desc, 64)
        monkeypatch.setenv("MSGPU_ED_DP", "1")
        dp = st.edit_distance(da.data_ptr(), db.data_ptr(), desc, 64)
    assert list(got) == want
    assert list(dp) == want
