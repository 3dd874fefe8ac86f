#!/usr/bin/env python3
"""The `e2e` leg of bench.py alone (files in -> files out at the size of a workload), several times over:
    python tools/e2e_only.py [workload] [reps] [factor]        (MSGPU_PARSE_DEBUG=1 prints the loaders' own phases)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from muchsalsa_amd import synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    w = dict(bench.WORKLOADS[name])
    factor = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # (a job `factor` times the workload, same generator)
    w["n_reads"], w["n_anchors"] = w["n_reads"] * factor, w["n_anchors"] * factor
    tab = synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"])
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    for _ in range(reps):
        r = bench.e2e_leg(None, w, tab, None, threads)
        print(json.dumps({"wall_s": round(r["wall_s"], 4), "stage_s": r["stage_s"], "contigs": r["counts"]["contigs"],
                          "target_bases": r["counts"]["target_bases"], "input_bytes": r["input_bytes"]}), flush=True)


if __name__ == "__main__":
    main()
