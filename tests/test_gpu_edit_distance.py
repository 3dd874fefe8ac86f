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


@pytest.mark.parametrize("band", [127, 40, 8, 0])
def test_random_pairs_match_full_dp(oracle, band):
    from muchsalsa_amd import sequences as S
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
