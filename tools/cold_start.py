#!/usr/bin/env python3
"""What a fresh process pays before its first kernel runs: library load, HIP runtime, first context, first launch.
    python tools/cold_start.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import numpy as np  # noqa: E402
t1 = time.perf_counter()
from muchsalsa_amd import _lib, overlap, synth  # noqa: E402
t2 = time.perf_counter()
L = _lib.lib()
t3 = time.perf_counter()
ctx = overlap.OverlapContext(0)
t4 = time.perf_counter()
rows, rn, an = synth.accepted_rows(synth.paf_table(2000, 5000, 10000, 7))
t5 = time.perf_counter()
ctx.overlap_batched(rows, 0, resident=True, edgematches=False)
t6 = time.perf_counter()
ctx.overlap_batched(rows, 0, resident=True, edgematches=False)
t7 = time.perf_counter()
print("numpy %.0f ms, package %.0f ms, dlopen libmsgpu %.0f ms, first context (HIP runtime) %.0f ms, first call %.0f ms, second call %.1f ms"
      % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t6 - t5), 1e3 * (t7 - t6)))
