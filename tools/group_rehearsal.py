#!/usr/bin/env python3
"""msgpu_group_overlap with N members at the size of a BASELINE.json configuration, checked: the merged edge list must be the
host statement of the merge over the N shard tables that N single contexts (msgpu_set_shard) produce, byte for byte, and the
EdgeMatch counts must add up.

    python tools/group_rehearsal.py [members = 8] [workload = cfg3]
        N members on ONE GPU (MSGPU_GROUP_TRANSPORT=copy: the all-gathers carried by device-to-device copies, every other step
        the code RCCL would drive).  A rehearsal of the n > 1 path of the C++ group, not a scaling measurement.
    python tools/group_rehearsal.py --devices 0,1,2,3,4,5,6,7 [--workload cfg3] [--json]
        one member per listed device, the all-gathers through RCCL over xGMI: the path as a libms caller runs it on a node.
        With --json the last line of stdout is one JSON object (bench.py --gpus N attaches it to its line as "group_on_node")."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("members", nargs="?", type=int, default=8)
    ap.add_argument("workload_pos", nargs="?", default=None)
    ap.add_argument("--devices", default=None, help="comma-separated device ordinals: one member each, RCCL transport")
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    on_node = args.devices is not None
    devices = [int(d) for d in args.devices.split(",")] if on_node else [0] * args.members
    if not on_node:
        os.environ["MSGPU_GROUP_TRANSPORT"] = "copy"
    from bench import WORKLOADS
    from muchsalsa_amd import distributed as D, overlap, synth
    n = len(devices)
    w = WORKLOADS[args.workload_pos or args.workload]
    rows, rn, an = synth.accepted_rows(synth.paf_table(w["n_reads"], w["read_len"], w["n_anchors"], w["seed"]))
    pinned = overlap.PinnedRows(rows)
    shards, n_ems = [], 0
    for r in range(n):
        with overlap.OverlapContext(devices[r]) as ctx:
            ctx.set_shard(r, n)
            ctx.load_rows(pinned.array)
            ctx.calculate_edges()
            ctx.chaining_and_overlaps()
            t = ctx.tables()
            n_ems += len(t["ems"])
            shards.append({k: t[k] for k in ("edges", "orders", "ids")})
        print("shard %d of %d (device %d): %d edges, %d orders" % (r, n, devices[r], len(shards[-1]["edges"]), len(shards[-1]["orders"])),
              file=sys.stderr if args.json else sys.stdout, flush=True)
    want = D.merge_tables_host(shards)
    with overlap.OverlapGroup(devices) as grp:
        t0 = time.perf_counter()
        grp.overlap(pinned, copy=False)  # communicators, buffers
        first_ms = 1e3 * (time.perf_counter() - t0)
        best = None
        for _ in range(max(1, args.reps)):
            got, info = grp.overlap(pinned, copy=False)
            for k in ("edges", "orders", "ids"):
                assert got[k].tobytes() == want[k].tobytes(), k
            assert info["n_ems"] == n_ems and info["n_members"] == n
            if best is None or info["wall_ms"] < best["wall_ms"]:
                best = dict(info, n_edges=int(len(got["edges"])), n_orders=int(len(got["orders"])), n_ids=int(len(got["ids"])))
    pinned.close()
    if args.json:
        print(json.dumps({
            "members": n, "devices": devices, "transport": "rccl" if on_node else "copy (rehearsal on shared devices: not a measurement)",
            "workload": w["name"], "wall_ms": best["wall_ms"], "compute_ms": best["compute_ms"], "exchange_ms": best["exchange_ms"],
            "first_call_ms": first_ms, "overlap_pairs_per_s": best["n_edges"] / (best["wall_ms"] * 1e-3), "edges": best["n_edges"],
            "slab_bytes": int(best["slab_bytes"]), "id_bytes": int(best["id_bytes"]), "rows_sliced": bool(best["rows_sliced"]),
            "verified": "merged edge / order / id tables == the host merge of the %d single-context shard tables, byte for byte, "
                        "on every repetition" % n,
            "stage": "msgpu_group_overlap (C++, one process, a host thread per member): rows in pinned host memory -> 1/n per link + "
                     "in-place all-gather -> index on every member -> shard v1 %% n -> msgpu_pack_wire -> ONE grouped all-gather -> "
                     "msgpu_merge_wire -> merged tables in host memory as n slices; wall = host clock of the best of %d calls"
                     % max(1, args.reps)}), flush=True)
    else:
        print("group of %d members%s: merged list of %d edges, %d orders, %d ids == the host merge of the %d shard tables, byte for "
              "byte; slab %d bytes per member (%d-byte ids); wall %.1f ms, compute %.1f, exchange %.2f%s"
              % (n, "" if on_node else " on one GPU (copy transport)", best["n_edges"], best["n_orders"], best["n_ids"], n,
                 best["slab_bytes"], best["id_bytes"], best["wall_ms"], best["compute_ms"], best["exchange_ms"],
                 "" if on_node else " (members share the GPU and its link: not a measurement)"), flush=True)


if __name__ == "__main__":
    main()
