#!/usr/bin/env python3
"""What the FIRST msgpu_overlap_batched_ex call of a context costs beyond a steady-state one (allocations of device and
page-locked tables), on BASELINE configs[2]:   python tools/first_call_timing.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    w = bench.WORKLOADS["cfg3"]
    rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
    pinned = overlap.PinnedRows(rows)
    with overlap.OverlapContext(0) as warm:  # (the HIP runtime, the code object)
        warm.overlap_batched(pinned.array[:1000], 0, copy=False, resident=True, edgematches=False)
    for rep in range(3):
        t0 = time.perf_counter()
        ctx = overlap.OverlapContext(0)
        ctx.set_id_space(len(rn), len(an))
        t1 = time.perf_counter()
        ctx.overlap_batched(pinned.array, 0, copy=False, resident=True, edgematches=False)
        t2 = time.perf_counter()
        ctx.overlap_batched(pinned.array, 0, copy=False, resident=True, edgematches=False)
        t3 = time.perf_counter()
        ctx.close()
        t4 = time.perf_counter()
        print("create %.1f ms, first call %.1f ms, second call %.1f ms, close %.1f ms" % (
            1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)


if __name__ == "__main__":
    main()
