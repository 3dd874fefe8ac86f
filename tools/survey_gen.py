#!/usr/bin/env python3
"""Synthetic PAF generator with the draw order SURVEY.md section 8(d) describes (CPython `random`, MT19937).

SURVEY.md records aggregate counts that the survey measured with the real reference on this generator's
cfg2 output (R, A, P_eval, P_emit, E, C, O, shadow edges, sum|ids|).  Re-creating the same input lets
tests/test_oracle_survey_counts.py check the CPU oracle against those recorded reference aggregates.
It is a cross-check, not a pin: the aggregates are counts, not tables.

usage: survey_gen.py READS READ_LEN ANCHORS GENOME SEED OUT.paf
"""
import bisect
import random
import sys


def generate(n_reads, read_len, n_anchors, genome, seed):
    rng = random.Random(seed)
    anchors = []
    for _ in range(n_anchors):
        length = rng.randint(500, 1500)
        anchors.append((rng.randint(0, genome - length), length))
    by_start = sorted(range(n_anchors), key=lambda a: anchors[a][0])
    starts = [anchors[a][0] for a in by_start]
    reads = []
    for _ in range(n_reads):
        start = rng.randint(0, genome - read_len)
        reads.append((start, rng.random() < 0.5))
    per_anchor = {}
    for r, (r_start, fwd) in enumerate(reads):
        lo = bisect.bisect_left(starts, r_start - 1500)
        hi = bisect.bisect_right(starts, r_start + read_len)
        for k in range(lo, hi):
            a = by_start[k]
            a_start, a_len = anchors[a]
            g_lo = max(a_start, r_start)
            g_hi = min(a_start + a_len, r_start + read_len)
            if g_hi - g_lo < 420:
                continue
            q_lo, q_hi = g_lo - a_start, g_hi - a_start
            if fwd:
                t_lo, t_hi = g_lo - r_start, g_hi - r_start
            else:
                t_lo, t_hi = r_start + read_len - g_hi, r_start + read_len - g_lo
            t_lo = max(0, t_lo + rng.randint(-15, 15))
            t_hi = min(read_len, t_hi + rng.randint(-15, 15))
            n_match = int((q_hi - q_lo) * rng.uniform(0.86, 0.97))
            per_anchor.setdefault(a, []).append(
                "u%d\t%d\t%d\t%d\t%s\tr%d\t%d\t%d\t%d\t%d\t%d\t60"
                % (a, a_len, q_lo, q_hi, "+" if fwd else "-", r, read_len, t_lo, t_hi, n_match, q_hi - q_lo))
    lines = []
    for a in range(n_anchors):
        lines.extend(per_anchor.get(a, ()))
    # trailing sentinel: BlastFileReader.cpp:76 never parses the last line
    lines.append("u0\t1\t0\t1\t+\tr0\t1\t0\t1\t0\t1\t0")
    return lines


def main(argv):
    n_reads, read_len, n_anchors, genome, seed = (int(x) for x in argv[1:6])
    lines = generate(n_reads, read_len, n_anchors, genome, seed)
    with open(argv[6], "w") as f:
        f.write("\n".join(lines))
        f.write("\n")
    print("rows", len(lines) - 1)


if __name__ == "__main__":
    main(sys.argv)
