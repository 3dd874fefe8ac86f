#!/usr/bin/env python3
"""The executable as a user starts it: `python -m muchsalsa_amd contigs.paf unitigs.fa nanopore.fa outdir` as a fresh
process on BASELINE configs[2] written to tmpfs -- wall time of the whole process (interpreter start, library load, HIP
runtime, the run, exit) next to the seconds the run reports for itself.   python tools/cli_cold.py [reps]"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from muchsalsa_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    w = bench.WORKLOADS["cfg3"]
    tab = synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"])
    d = tempfile.mkdtemp(prefix="msgpu_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        bench.write_e2e_inputs(d, w, tab)
        for rep in range(reps):
            out = os.path.join(d, "out%d" % rep)
            os.mkdir(out)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "muchsalsa_amd", os.path.join(d, "contigs.paf"), os.path.join(d, "unitigs.fa"),
                                os.path.join(d, "nanopore.fa"), out], cwd=ROOT, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                print(r.stderr[-2000:])
                raise SystemExit(r.returncode)
            res = json.loads(r.stdout.strip().splitlines()[-1])
            sec = res["seconds"]
            inside = sum(sec[k] for k in ("parse_paf", "overlap_gpu", "contraction_gpu", "graph_host", "path_edgematches", "sequences_wait",
                                          "assemble", "write", "collect", "teardown") if k in sec)
            print("process wall %.3f s; stages on the run's critical path %.3f s; device_init %.3f s; contigs %d" % (
                wall, inside, sec.get("device_init", 0.0), res["contigs"]), flush=True)
            if rep == reps - 1:
                print(json.dumps(sec))
        exe = os.path.join(ROOT, "muchsalsa_amd", "muchsalsa_gpu")  # the same without Python in the process
        for rep in range(reps if os.path.exists(exe) else 0):
            out = os.path.join(d, "cpp%d" % rep)
            os.mkdir(out)
            t0 = time.perf_counter()
            r = subprocess.run([exe, os.path.join(d, "contigs.paf"), os.path.join(d, "unitigs.fa"), os.path.join(d, "nanopore.fa"), out],
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                print(r.stderr[-2000:])
                raise SystemExit(r.returncode)
            print("muchsalsa_gpu (C++): process wall %.3f s; %s" % (wall, r.stdout.strip().splitlines()[-2]), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
