#!/usr/bin/env python3
"""Per-kernel averages (per dispatch) of the counters collected by tools/pmc_passes.sh -> OUTDIR/pmc_summary.csv, plus
OUTDIR/pmc_meta.json naming the (workload, world) the capture belongs to.  bench.py reads the newest such pair under
profiles/ for `roofline.traffic` (HBM bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per MI355X_MICROARCH.md: on
gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads; both counters are in KiB).

    python tools/pmc_summarize.py OUTDIR [workload] [world]
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)  # argument list
    return name.strip()


def main():
    out = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    acc, counters = {}, []
    for f in sorted(glob.glob(os.path.join(out, "pmc_*_counter_collection.csv")) +
                    glob.glob(os.path.join(out, "**", "pmc_*_counter_collection.csv"), recursive=True)):
        per_run = {}
        for row in csv.DictReader(open(f)):
            k, c = short(row["Kernel_Name"]), row["Counter_Name"]
            if not k.startswith(("msgpu::", "void msgpu::")):
                continue
            d = per_run.setdefault((k, c), [0.0, set()])
            d[0] += float(row["Counter_Value"])
            d[1].add(row["Dispatch_Id"])
            if c not in counters:
                counters.append(c)
        for (k, c), (tot, disp) in per_run.items():
            if c in acc.get(k, {}) and "gather" in os.path.basename(f) and "k_gather" not in k:
                continue  # the gather passes are there for the gather kernels; the other kernels keep their own passes
            acc.setdefault(k, {})[c] = (tot / max(len(disp), 1), len(disp))
    with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
        f.write("kernel,dispatches," + ",".join(counters) + "\n")
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("FETCH_SIZE", (0, 0)))[0]):
            n = max(v[1] for v in acc[k].values())
            f.write('"%s",%d,' % (k, n) + ",".join("%g" % acc[k][c][0] if c in acc[k] else "" for c in counters) + "\n")
    with open(os.path.join(out, "pmc_meta.json"), "w") as f:
        json.dump({"workload": workload, "world": world,
                   "command": "rocprofv3 --pmc <one counter set per pass> -- python3 bench.py --steps 3 --warmup 1 "
                              "--cpu-sample-reads 0 --kernels-only (tools/pmc_passes.sh)",
                   "units": "per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB"}, f, indent=1)
        f.write("\n")
    print("wrote", os.path.join(out, "pmc_summary.csv"), "kernels:", len(acc), "counters:", counters)


if __name__ == "__main__":
    main()
