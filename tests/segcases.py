"""Random inputs for the segment builders of assemblePath (shared by the CPU and GPU tests)."""
import numpy as np

from muchsalsa_amd.synth import ROW_DTYPE


def make_world(seed, n_reads=12, n_unitigs=10):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTACGTACGTN", dtype=np.uint8)
    reads = [bytes(rng.choice(alpha, int(n))) for n in rng.integers(3000, 9000, n_reads)]
    unis = [bytes(rng.choice(alpha, int(n))) for n in rng.integers(600, 1600, n_unitigs)]
    return rng, reads, unis


def random_match(rng, reads, unis, read_id=None):
    rid = int(rng.integers(0, len(reads))) if read_id is None else read_id
    aid = int(rng.integers(0, len(unis)))
    ul, rl = len(unis[aid]), len(reads[rid])
    i_lo = int(rng.integers(0, ul - 450))
    i_hi = int(rng.integers(i_lo + 420, ul))
    n_lo = int(rng.integers(0, rl - 600))
    n_hi = int(min(rl - 1, n_lo + (i_hi - i_lo) + rng.integers(-30, 31)))
    m = np.zeros((), dtype=ROW_DTYPE)
    m["anchor_id"], m["read_id"], m["read_len"] = aid, rid, rl
    m["i_lo"], m["i_hi"], m["n_lo"], m["n_hi"] = i_lo, i_hi, n_lo, n_hi
    m["score"], m["line"], m["flags"] = 500, 0, int(rng.integers(0, 2))
    ov_lo = int(rng.integers(i_lo, i_lo + 150))
    ov_hi = int(rng.integers(max(ov_lo, i_hi - 150), i_hi + 1))
    return m, (ov_lo, ov_hi)


def apply_pieces(pieces, stores, revcomp):
    """Host statement of what the gather kernel does with a piece list (test helper)."""
    out = bytearray(int(max((int(p["dst_off"]) + int(p["len"]) for p in pieces), default=0)))
    for p in pieces:
        src = stores[1 if int(p["flags"]) & 1 else 0]
        s = src[int(p["src_off"]): int(p["src_off"]) + int(p["len"])]
        if int(p["flags"]) & 2:
            s = revcomp(s)
        out[int(p["dst_off"]): int(p["dst_off"]) + len(s)] = s
    return bytes(out)
