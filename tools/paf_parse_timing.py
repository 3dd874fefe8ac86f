#!/usr/bin/env python3
"""msgpu_parse_paf alone on the machine: BASELINE configs[2] as PAF text (270 MB) in tmpfs, parsed several times
(MSGPU_PARSE_DEBUG=1 prints the loader's phases).   python tools/paf_parse_timing.py [reps] [factor]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import bench  # noqa: E402
from muchsalsa_amd import overlap, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    w = dict(bench.WORKLOADS["cfg3"])
    factor = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # (a file `factor` times as large, same generator)
    w["n_reads"], w["n_anchors"] = w["n_reads"] * factor, w["n_anchors"] * factor
    tab = synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"])
    d = tempfile.mkdtemp(prefix="msgpu_paf_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        n = len(tab["qname_id"])
        path = os.path.join(d, "contigs.paf")
        pd.DataFrame({"q": tab["qname_id"], "ql": tab["qlen"], "qs": tab["qstart"], "qe": tab["qend"],
                      "s": np.where(tab["strand"], "+", "-"), "t": tab["tname_id"], "tl": tab["tlen"], "ts": tab["tstart"],
                      "te": tab["tend"], "nm": tab["nmatch"], "bl": tab["qend"] - tab["qstart"],
                      "mq": np.full(n, 60)}).to_csv(path, sep="\t", header=False, index=False)
        with open(path, "a") as f:
            f.write("0\t1\t0\t1\t+\t0\t1\t0\t1\t0\t1\t0\n")
        params = overlap.default_params()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            p = overlap.parse_paf(path, params)
            times.append(time.perf_counter() - t0)
            rows = len(p.rows)
            del p
        print("rows %d; msgpu_parse_paf %s ms (min %.1f)" % (rows, " ".join("%.1f" % (1e3 * t) for t in times), 1e3 * min(times)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
