"""PipelinedExchange on the GPU as a run uses it -- RCCL ("nccl") at world 1, a communication thread, a communication
stream -- with a rank that outgrows the slab capacity in two consecutive batches (the second goes out before the first one's
headers were read).  Every batch's merged tables must be the rank's own tables, byte for byte; the slab size every batch
went out with must be the one the unthreaded protocol gives.  Prints one JSON line; started by tests/test_gpu_parity.py."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE))]


def main():
    import torch
    import torch.distributed as dist
    from muchsalsa_amd import distributed as D, overlap, synth
    from muchsalsa_amd._lib import EDGE_DTYPE, ORDER_DTYPE
    wire = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        small = synth.synth_rows(600, 5000, 2400, 21)
        large = synth.synth_rows(1800, 5000, 7200, 22)   # about three times the tables
        work = torch.cuda.Stream(device=dev)
        ctx = overlap.OverlapContext(0)
        ctx.set_stream(work.cuda_stream)
        results, sizes = [], []

        def fill(slab, offs):
            ctx.pack_wire(slab.data_ptr() + offs[0], slab.data_ptr() + offs[1], slab.data_ptr() + offs[2], id_bytes=wire)

        def merge(gathered, allc, offs, slab_bytes, k, stream):
            tot = allc.sum(axis=0)
            out = [torch.empty(max(int(n), 1) * sz, dtype=torch.uint8, device=dev)
                   for n, sz in zip(tot, (EDGE_DTYPE.itemsize, ORDER_DTYPE.itemsize, 4))]
            # (torch.empty queues nothing; the merge runs on the communication stream `stream`, behind the all-gather)
            ctx.merge_wire(gathered.data_ptr(), allc, slab_bytes, offs, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                           id_bytes=wire, stream=stream.cuda_stream)
            results.append((out, tot))
            sizes.append(int(slab_bytes))

        def run(threaded):
            results.clear()
            sizes.clear()
            own = []
            pe = D.PipelinedExchange(dev, merge, wire=wire, threaded=threaded)
            with torch.cuda.stream(work):
                for b in range(6):
                    rows = large if b in (2, 3) else small
                    ctx.load_rows(rows)
                    ctx.calculate_edges()
                    ctx.chaining_and_overlaps()
                    c = ctx.counts()
                    own.append(ctx.tables())
                    pe.submit((c.n_edges, c.n_orders, c.n_ids), fill)
                    pe.collect()
                pe.drain()
            pe.close()
            torch.cuda.synchronize()
            assert len(results) == 6
            for b, ((out, tot), t) in enumerate(zip(results, own)):
                assert tuple(int(x) for x in tot) == (len(t["edges"]), len(t["orders"]), len(t["ids"])), b
                for name, o in zip(("edges", "orders", "ids"), out):
                    want = t[name].view(np.uint8).reshape(-1)
                    if name == "edges":  # em_off is rank-local by contract and the same here (one rank)
                        pass
                    assert o.cpu().numpy()[: len(want)].tobytes() == want.tobytes(), (threaded, b, name)
            return list(sizes), pe.regrows, pe.collectives

        plain = run(False)
        threaded = run(True)
        assert plain == threaded, (plain, threaded)
        assert plain[1] == 2 and plain[2] == 1 + 6 + 2, plain
        assert plain[0][1] == plain[0][0] and plain[0][3] > plain[0][0], plain[0]  # batch 3 went out with the OLD capacity and was re-laid
        print(json.dumps({"ok": True, "slab_bytes": plain[0], "regrows": plain[1], "collectives": plain[2], "wire": wire}))
        ctx.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
