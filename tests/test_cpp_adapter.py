"""include/msgpu_adapter.hpp: compiles against mocks of the reference interface; error behaviour; GPU end-to-end."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def adapter_bin(tmp_path_factory):
    import __graft_entry__ as g
    g.build()
    out = str(tmp_path_factory.mktemp("cpp") / "test_adapter")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp"), "-o", out,
                    "-pthread", "-L", os.path.join(ROOT, "muchsalsa_amd"), "-lmsgpu",
                    "-Wl,-rpath," + os.path.join(ROOT, "muchsalsa_amd"), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def test_adapter_compiles_and_reports_errors_like_the_reference(adapter_bin):
    r = subprocess.run([adapter_bin, "--errors"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() in ("ok gpu", "ok nogpu")


@pytest.mark.gpu
def test_adapter_end_to_end_fills_reference_shaped_objects(adapter_bin, oracle, tmp_path):
    from muchsalsa_amd import synth
    tab = synth.paf_table(300, 4000, 1000, 31)
    path = tmp_path / "in.paf"
    path.write_text("\n".join(synth.paf_lines(tab)) + "\n")
    dump = tmp_path / "objects.txt"
    r = subprocess.run([adapter_bin, str(path), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    rows = oracle.parse_paf(str(path))["rows"]
    t = oracle.overlap(rows)
    assert got == {"vertices": int(rows["read_id"].max()) + 1, "edges": len(t["edges"]),
                   "vertexmatches": t["rows_alive"], "edgematches": len(t["ems"]), "orders": len(t["orders"]),
                   "shadows": t["shadow_edges"], "ids": len(t["ids"])}
    # ... and the OBJECTS themselves: every Vertex, VertexMatch, Edge (shadow flag), EdgeMatch and EdgeOrder (vertices, offsets
    # bit for bit, score, flags, ids) that fillReferenceObjects put into the reference-shaped containers, against the oracle's
    # tables.  (The test's PAF names reads and anchors in id order, so the mock registries number them as the loader did.)
    import struct
    bits = lambda d: "%016x" % struct.unpack("<Q", struct.pack("<d", float(d)))[0]  # noqa: E731
    want = []
    for v in range(len(t["read_len"])):
        want.append("V %d %d %d" % (v, t["read_len"][v], t["read_first_line"][v]))
    best = {}
    for r_ in rows:  # lowest line per (read, anchor): MatchMap.cpp:64-80
        k = (int(r_["read_id"]), int(r_["anchor_id"]))
        if k not in best or int(best[k]["line"]) > int(r_["line"]):
            best[k] = r_
    for (rd, an), r_ in sorted(best.items()):
        rr = float(int(r_["i_hi"]) - int(r_["i_lo"]) + 1) / float(int(r_["n_hi"]) - int(r_["n_lo"]) + 1)
        want.append("VM %d %d %d %d %d %d %s %d %d %d %d" % (rd, an, r_["n_lo"], r_["n_hi"], r_["i_lo"], r_["i_hi"], bits(rr),
                                                            int(r_["flags"]) & 1, r_["score"], (int(r_["flags"]) >> 1) & 1, r_["line"]))
    e, em, o, ids = t["edges"], t["ems"], t["orders"], t["ids"]
    for i in range(len(e)):
        v1, v2 = int(e["v1"][i]), int(e["v2"][i])
        want.append("E %d %d %d %d" % (v1, v2, e["shadow"][i], e["order_cnt"][i]))
        for m in em[int(e["em_off"][i]): int(e["em_off"][i]) + int(e["em_cnt"][i])]:
            want.append("EM %d %d %d %d %d %d %s %d %d" % (v1, v2, m["anchor_id"], m["ov_lo"], m["ov_hi"], int(m["flags"]) & 1,
                                                          bits(m["score"]), (int(m["flags"]) >> 1) & 1, m["line"]))
        for k, q in enumerate(o[int(e["order_off"][i]): int(e["order_off"][i]) + int(e["order_cnt"][i])]):
            fl = int(q["flags"])
            want.append(("O %d %d %d %d %d %d %s %s %d %d %d %d" % (v1, v2, k, q["start"], q["end"], q["base"], bits(q["left_offset"]),
                                                                   bits(q["right_offset"]), (fl >> 1) & 1, q["score"], (fl >> 2) & 1,
                                                                   (fl >> 3) & 1)) +
                        "".join(" %d" % x for x in ids[int(q["ids_off"]): int(q["ids_off"]) + int(q["ids_cnt"])]))
    have = dump.read_text().splitlines()
    assert len(have) == len(want) and sorted(have) == sorted(want)
    # EdgeMatches and EdgeOrders keep their order inside an edge (vStart order; minus paths first): compare those in sequence
    seq = lambda lines, tag: [ln for ln in lines if ln.startswith(tag)]  # noqa: E731
    assert seq(have, "EM ") == seq(want, "EM ") and seq(have, "O ") == seq(want, "O ")


@pytest.mark.gpu
def test_cpp_whole_flow_writes_the_same_files_as_the_python_driver(adapter_bin, tmp_path):
    """msgpu::assemble (the C++ statement of main(), include/msgpu_adapter.hpp) and muchsalsa_amd.pipeline.run drive the
    same C-ABI: identical temp_1.* files from the same inputs, with no Python between the calls."""
    from graphcases import make_dataset
    from muchsalsa_amd import pipeline
    make_dataset(tmp_path, 7, 8, True)
    (tmp_path / "cpp").mkdir()
    (tmp_path / "py").mkdir()
    r = subprocess.run([adapter_bin, "--assemble", str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"),
                        str(tmp_path / "nanopore.fq"), str(tmp_path / "cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    want = pipeline.run(str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / "nanopore.fq"),
                        str(tmp_path / "py"), threads=2)
    for k in ("rows", "edges", "contraction_edges", "paths", "paths_skipped", "contigs", "target_bases", "queries"):
        assert got[k] == want[k], k
    assert got["contigs"] >= 1 and got["target_bases"] > 200_000
    for name in ("temp_1.target.fa", "temp_1.query.fa", "temp_1.align.paf"):
        assert (tmp_path / "cpp" / name).read_bytes() == (tmp_path / "py" / name).read_bytes(), name


@pytest.mark.gpu
def test_executable_takes_the_reference_argument_list(tmp_path):
    """muchsalsa_amd/muchsalsa_gpu (csrc/muchsalsa_main.cpp, built by the library's Makefile): the reference's executable on
    libmsgpu with no Python in the process -- same argument list (src/Application.cpp:34-39), same three files as the Python
    driver writes for the same inputs, "Finished assembly" at the end; too few arguments -> usage and -1."""
    import __graft_entry__ as g
    from graphcases import make_dataset
    from muchsalsa_amd import pipeline
    g.build()
    exe = os.path.join(ROOT, "muchsalsa_amd", "muchsalsa_gpu")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "only", "three", "args"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and "usage" in r.stderr
    make_dataset(tmp_path, 12, 5, False)
    (tmp_path / "cpp").mkdir()
    (tmp_path / "py").mkdir()
    r = subprocess.run([exe, str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / "nanopore.fa"),
                        str(tmp_path / "cpp"), "3", "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "Finished assembly"
    got = json.loads(lines[-2])
    want = pipeline.run(str(tmp_path / "contigs.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / "nanopore.fa"),
                        str(tmp_path / "py"), threads=3)
    for k in ("rows", "edges", "contraction_edges", "paths", "paths_skipped", "contigs", "target_bases", "queries"):
        assert got[k] == want[k], k
    for name in ("temp_1.target.fa", "temp_1.query.fa", "temp_1.align.paf"):
        assert (tmp_path / "cpp" / name).read_bytes() == (tmp_path / "py" / name).read_bytes(), name
    r = subprocess.run([exe, str(tmp_path / "missing.paf"), str(tmp_path / "unitigs.fa"), str(tmp_path / "nanopore.fa"),
                        str(tmp_path / "cpp")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "muchsalsa_gpu:" in r.stderr
