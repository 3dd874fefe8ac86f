#!/usr/bin/env python3
"""msgpu_group_overlap with N members on ONE GPU at the size of BASELINE.json configs[2] (MSGPU_GROUP_TRANSPORT=copy: the
all-gather carried by device-to-device copies, every other step the code RCCL would drive): the merged edge list must be the
host statement of the merge over the N shard tables that N single contexts (msgpu_set_shard) produce, byte for byte, and the
EdgeMatch counts must add up.  A rehearsal of the n > 1 path of the C++ group, not a scaling measurement.
    python tools/group_rehearsal.py [members = 8] [workload = cfg3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MSGPU_GROUP_TRANSPORT"] = "copy"
import numpy as np  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from muchsalsa_amd import distributed as D, overlap, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    w = WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "cfg3"]
    rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
    pinned = overlap.PinnedRows(rows)
    shards, n_ems = [], 0
    for r in range(n):
        with overlap.OverlapContext(0) as ctx:
            ctx.set_shard(r, n)
            ctx.load_rows(pinned.array)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            t = ctx.tables()
            n_ems += len(t["ems"])
            shards.append({k: t[k] for k in ("edges", "orders", "ids")})
        print("shard %d of %d: %d edges, %d orders" % (r, n, len(shards[-1]["edges"]), len(shards[-1]["orders"])), flush=True)
    want = D.merge_tables_host(shards)
    with overlap.OverlapGroup([0] * n) as grp:
        grp.overlap(pinned, copy=False)
        t0 = time.perf_counter()
        got, info = grp.overlap(pinned, copy=False)
        dt = time.perf_counter() - t0
        for k in ("edges", "orders", "ids"):
            assert got[k].tobytes() == want[k].tobytes(), k
        assert info["n_ems"] == n_ems and info["n_members"] == n
        print("group of %d members on one GPU (copy transport): merged list of %d edges, %d orders, %d ids == the host merge of the %d "
              "shard tables, byte for byte; slab %d bytes per member (%d-byte ids); wall %.1f ms (members share the GPU and its link: "
              "not a measurement)" % (n, len(got["edges"]), len(got["orders"]), len(got["ids"]), n, info["slab_bytes"], info["id_bytes"],
                                      1e3 * dt), flush=True)


if __name__ == "__main__":
    main()
