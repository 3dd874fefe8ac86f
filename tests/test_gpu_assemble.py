"""assemblePath end to end on the GPU: host layout + one gather launch + FASTA wrapping kernel, compared byte for byte
with the texts the Python restatement of ap.cpp writes (temp_1.target.fa / temp_1.query.fa / temp_1.align.paf)."""
import numpy as np
import pytest

from asmcases import World
from test_assemble_path import fuzz_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["bytes", "two_bit"])
def gpu_world(request, oracle, tmp_path_factory):
    from muchsalsa_amd.overlap import build_overlaps
    from muchsalsa_amd.sequences import ILLUMINA, NANOPORE, SeqFile, SeqStore
    w = World(300, 5000, 1500, 7, jitter=15)
    w.attach(build_overlaps(w.rows))  # the overlap tables come from the HIP path
    d = tmp_path_factory.mktemp("asmgpu")
    for name, seqs in (("n.fa", w.nano), ("i.fa", w.illu)):
        with open(d / name, "wb") as f:
            for i in range(len(seqs)):
                f.write(b">s%d\n" % i + seqs[i] + b"\n")
    w.files = (SeqFile(str(d / "n.fa")), SeqFile(str(d / "i.fa")))
    w.store = SeqStore(device=0)
    w.store.upload(NANOPORE, w.files[0])
    w.store.upload(ILLUMINA, w.files[1])
    if request.param == "two_bit":
        w.store.pack()
    return w


def test_texts_match_oracle(gpu_world, oracle):
    from oracle.ms_assemble_py import AssemblyError, assemble_path
    from muchsalsa_amd.assembly import Assembly
    from muchsalsa_amd.overlap import MsgpuError
    w = gpu_world
    asm, want = Assembly(w.store), []
    order = np.argsort(w.read_start)
    k = 0
    for s in order[:80:4]:  # plain chains, both walking directions
        for flip in (False, True):
            path, steps = w.chain(int(s), max_len=10, flip_all=flip)
            if len(path) >= 2:
                want.append(assemble_path(path, steps, w.vm, {}, w.nano, w.illu, k))
                asm.add_path(path, steps, w.rows, None, k)
                k += 1
    rng = np.random.default_rng(11)
    skipped = 0
    for trial in range(200):  # fuzzed inputs (every branch of the layout)
        case = fuzz_case(w, rng, trial)
        if case is None:
            continue
        path, steps, rows, vm, contains = case
        ocont = {r: [dict(nano=c["nano"], dir=c["dir"], matches=c["matches"]) for c in v] for r, v in contains.items()}
        pcont = {r: [dict(nano=c["nano"], dir=c["dir"], anchors=list(c["matches"])) for c in v]
                 for r, v in contains.items()}
        try:
            r = assemble_path(path, steps, vm, ocont, w.nano, w.illu, k)
        except (AssemblyError, KeyError, IndexError):
            with pytest.raises(MsgpuError):
                asm.add_path(path, steps, rows, pcont, k)
            skipped += 1
            continue
        asm.add_path(path, steps, rows, pcont, k)
        want.append(r)
        k += 1
    assert len(want) > 120 and skipped > 5
    asm.finish()
    assert asm.text(2) == b"".join(r["paf"] for r in want)
    tgt, qry = asm.text(0), asm.text(1)
    assert tgt == b"".join(r["target_fa"] for r in want)
    assert qry == b"".join(r["query_fa"] for r in want)
    assert sum(len(r["target"]) for r in want) == int(asm.paths["target_len"].sum())
    # self-check on the device: every query against the stretch of the contig its PAF line names (banded DP kernel)
    band = 100
    dist, cells = asm.validate(band)
    k, n_in_band = 0, 0
    for r in want:
        tlen = len(r["target"])
        for name, seq, lb, rb in r["queries"]:
            lo, hi = max(lb, 0), min(rb, tlen - 1)
            window = r["target"][lo:hi + 1] if hi >= lo else b""
            assert int(dist[k]) == oracle.edit_distance_banded(seq, window, band), (name, lb, rb)
            n_in_band += int(dist[k]) <= band
            k += 1
    assert k == len(dist) and cells > 0
    assert n_in_band > 0.5 * k  # the query records do lie where their PAF lines say


def test_fasta_format_line_boundaries():
    """limitLength (ap.cpp:61-76) on the device for every length around the 60-column and 16-byte/1-KiB boundaries."""
    import ctypes as C

    import torch
    from oracle.ms_assemble_py import limit_length
    from muchsalsa_amd import _lib
    from muchsalsa_amd.sequences import SeqStore
    L = _lib.lib()
    store = SeqStore(device=0)
    rng = np.random.default_rng(3)
    lens = list(range(0, 200)) + [959, 960, 961, 1023, 1024, 1025, 6000, 6001, 65536 + 7]
    raw = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(sum(lens)) + 64)
    recs = np.zeros(len(lens), dtype=_lib.FASTA_RECORD_DTYPE)
    headers, want, roff, toff = b"", b"", 3, 5  # deliberately unaligned starts
    for i, n in enumerate(lens):
        h = b">rec.%d\n" % i
        recs[i] = (roff, toff, n, len(headers), len(h), 0)
        body = raw[roff:roff + n].tobytes()
        text = h + limit_length(body) + b"\n"
        assert len(text) == L.msgpu_fasta_text_bytes(len(h), n)
        want += text
        headers += h
        roff += n
        toff += len(text)
    d_raw = torch.from_numpy(raw).cuda()
    d_text = torch.full((toff + 32,), 0x2e, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = L.msgpu_fasta_format(store._h, C.c_void_p(d_raw.data_ptr()), recs.ctypes.data, len(recs), headers, len(headers),
                              C.c_void_p(d_text.data_ptr()), toff + 32, None)
    assert rc == 0
    got = d_text.cpu().numpy().tobytes()
    assert got[:5] == b"." * 5 and got[toff:] == b"." * 32  # nothing written outside the records
    assert got[5:toff] == want
    # capacity is checked before anything is launched
    assert L.msgpu_fasta_format(store._h, C.c_void_p(d_raw.data_ptr()), recs.ctypes.data, len(recs), headers,
                                len(headers), C.c_void_p(d_text.data_ptr()), toff - 1, None) == _lib.E_ARG


def test_seqctx_may_be_destroyed_before_its_assembly(gpu_world):
    """Contract of include/msgpu.h (msgpu_assembly_create): the assembly uses its msgpu_seqctx until finish / validate
    have returned, and msgpu_assembly_free never touches the context -- so the context may go first.  (Round 1 had a
    use-after-free exactly there: the assembly's free path gave buffers back through the dead context.)  Raw C-ABI
    calls, no Python wrapper bookkeeping; afterwards the GPU must still do unrelated work correctly."""
    import ctypes as C

    from muchsalsa_amd import _lib
    from muchsalsa_amd.assembly import Assembly
    from muchsalsa_amd.overlap import build_overlaps
    from muchsalsa_amd.sequences import ILLUMINA, NANOPORE
    w = gpu_world
    L = _lib.lib()
    for finish_first in (True, False):
        ctx = C.c_void_p()
        assert L.msgpu_seq_create(0, C.byref(ctx)) == 0
        for kind, f in ((NANOPORE, w.files[0]), (ILLUMINA, w.files[1])):
            assert L.msgpu_seq_upload(ctx, kind, f._h, None, len(f)) == 0
        asm = C.c_void_p()
        assert L.msgpu_assembly_create(ctx, C.byref(asm)) == 0
        path, steps = w.chain(int(np.argsort(w.read_start)[0]), max_len=8)
        prepared = Assembly.prepare(path, steps, w.rows, None, 0)
        assert L.msgpu_assembly_add_path(asm, C.byref(prepared[0])) == 0
        if finish_first:
            assert L.msgpu_assembly_finish(asm, None) == 0
            n = C.c_uint64()
            assert L.msgpu_assembly_text(asm, 0, C.byref(n)) and n.value > 0
        L.msgpu_seq_destroy(ctx)         # the context goes first ...
        L.msgpu_assembly_free(asm)       # ... and the assembly's free path must not reach into it
        t = build_overlaps(w.rows)       # the device is still healthy
        assert t["orders"].tobytes() == w.tables["orders"].tobytes()
