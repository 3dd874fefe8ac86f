#!/usr/bin/env python3
"""End-to-end run of muchsalsa_amd.pipeline on a generated data set of a chosen size (GPU box):
    python tools/e2e_scale.py [n_reads] [genome_len] [jitter]
Prints one JSON line: counts, per-stage seconds, and how much of the genome the contigs cover."""
import json
import os
import pathlib
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    genome_len = int(sys.argv[2]) if len(sys.argv) > 2 else n_reads * 625
    jitter = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    from graphcases import make_dataset
    from muchsalsa_amd import pipeline
    with tempfile.TemporaryDirectory() as tmp:
        d = pathlib.Path(tmp)
        t0 = time.perf_counter()
        rows, lay, genome, nano, illu, name = make_dataset(d, 1, jitter, False, n_reads, genome_len)
        t_gen = time.perf_counter() - t0
        (d / "out").mkdir()
        timings = {}
        t0 = time.perf_counter()
        res = pipeline.run(str(d / "contigs.paf"), str(d / "unitigs.fa"), str(d / name), str(d / "out"), timings=timings)
        res["wall_s"] = round(time.perf_counter() - t0, 3)
        res["seconds"] = {k: round(v, 4) for k, v in timings.items()}
        res["generator_s"] = round(t_gen, 2)
        res["genome"] = genome_len
        res["target_over_genome"] = round(res["target_bases"] / genome_len, 4)
        sizes = sorted((len(line) for line in (d / "out" / "temp_1.target.fa").read_bytes().split(b">")[1:]), reverse=True)
        res["largest_contig_text_bytes"] = sizes[:5]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
