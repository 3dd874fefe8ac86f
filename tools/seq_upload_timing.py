#!/usr/bin/env python3
"""Sequence files to HBM, the two ways, alone on the machine: msgpu_seq_parse + msgpu_seq_upload_bases against
msgpu_seq_parse_upload (MSGPU_PARSE_DEBUG=1 prints the parser's phases).   python tools/seq_upload_timing.py [reps]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from muchsalsa_amd import sequences as S, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    w = bench.WORKLOADS["cfg3"]
    n_reads, L, seed = w["n_reads"], w["read_len"], w["seed"]
    G, r_start, r_fwd = synth.read_layout(n_reads, L, seed)
    genome = synth.genome_bases(G, seed).tobytes()
    d = tempfile.mkdtemp(prefix="msgpu_seq_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        path = os.path.join(d, "nanopore.fa")
        with open(path, "wb") as f:
            for i in range(n_reads):
                f.write(b">%d\n" % i + genome[r_start[i]: r_start[i] + L] + b"\n")
        with S.SeqStore(0) as store:
            for rep in range(reps):
                t0 = time.perf_counter()
                f = S.SeqFile(path)
                t1 = time.perf_counter()
                store.upload_bases(S.NANOPORE, f)
                t2 = time.perf_counter()
                f.close()
                t3 = time.perf_counter()
                g = store.parse_upload(S.NANOPORE, path)
                t4 = time.perf_counter()
                g.close()
                print("parse %.1f ms + upload %.1f ms + free %.1f ms   |   parse_upload %.1f ms" % (
                    1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
