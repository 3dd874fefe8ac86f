"""The C-ABI library loads and exports exactly what include/msgpu.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from muchsalsa_amd import _lib
    return _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "msgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msgpu_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 24
    handle = lib.lib()
    bound = {name for name, _, _ in lib.SYMBOLS}
    for name in declared:
        assert hasattr(handle, name), "libmsgpu.so does not export %s" % name
        assert name in bound, "muchsalsa_amd._lib does not bind %s" % name
    assert bound == set(declared)


def test_record_layouts_match_oracle_header(lib):
    import ms_oracle_ctypes as O
    for a, b in ((lib.ROW_DTYPE, O.ROW_DTYPE), (lib.EDGE_DTYPE, O.EDGE_DTYPE), (lib.EM_DTYPE, O.EM_DTYPE),
                 (lib.ORDER_DTYPE, O.ORDER_DTYPE)):
        assert a == b
    assert C.sizeof(lib.Params) == 40 and C.sizeof(lib.Counts) == 88 and C.sizeof(lib.Timings) == 28


def test_default_params_are_the_reference_constants(lib):
    p = lib.Params()
    lib.lib().msgpu_default_params(C.byref(p))
    assert (p.min_matches, p.th_length, p.th_matches, p.th_overlap, p.wiggle_room, p.ratio_pct, p.alt_frac) == (
        400, 500, 500, 100, 300, 15.0, 0.75)


def test_no_cpu_fallback(lib):
    """Without a GPU msgpu_create must fail loudly; with one it must succeed."""
    import torch
    h = C.c_void_p()
    rc = lib.lib().msgpu_create(0, None, C.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        lib.lib().msgpu_destroy(h)
    else:
        assert rc == lib.E_NODEVICE
        from muchsalsa_amd import overlap
        with pytest.raises(overlap.MsgpuError):
            overlap.build_overlaps(np.zeros(0, dtype=lib.ROW_DTYPE))


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "muchsalsa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "ms_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)
    text = open(os.path.join(ROOT, "include", "msgpu.h")).read()
    assert "#include \"ms_oracle" not in text
