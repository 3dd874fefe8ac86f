#!/usr/bin/env python3
"""A1 loader throughput: msgpu_parse_paf on a synthetic PAF text (cfg2 by default), by host thread count."""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
from muchsalsa_amd import _lib, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
tab = synth.paf_table(**synth.CONFIGS[cfg])
path = os.path.join(tempfile.gettempdir(), "parse_bench_%s.paf" % cfg)
with open(path, "w") as f:
    lines = synth.paf_lines(tab)
    f.write("\n".join(lines) + "\n")
size = os.path.getsize(path) / 1e6
print("%s: %d lines, %.1f MB, %d host cpus" % (cfg, len(lines), size, os.cpu_count()))
L = _lib.lib()
for thr in (1, 2, 4, 8, 16):
    os.environ["MSGPU_PARSE_THREADS"] = str(thr)
    best = 1e9
    for _ in range(3):
        h = C.c_void_p()
        t = time.perf_counter()
        rc = L.msgpu_parse_paf(path.encode(), None, C.byref(h))
        dt = time.perf_counter() - t
        assert rc == 0
        L.msgpu_paf_free(h)
        best = min(best, dt)
    print("  %2d threads: %7.1f ms  %7.0f MB/s  %.2f M lines/s" % (thr, 1e3 * best, size / best, len(lines) / best / 1e6))
os.remove(path)
