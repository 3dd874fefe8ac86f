#!/usr/bin/env python3
"""Extract the golden strings of the reference's SequenceAccessor tests into a data fixture.

libms/tests/SA_test.cpp:11-136 checks whole-record fetches of test_data/fasta.fa and test_data/fastq.fq against
string literals.  This script (build container only: it reads /root/reference) copies the two data files and writes the
expected strings to tests/golden/ref_test_data/sa_test_expected.json.  Only DATA is copied -- no reference source.
"""
import json
import os
import re
import shutil

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_test_data")


def literals(text):
    """name -> value of every `const char *name = "...";` (with backslash-newline continuations)."""
    out = {}
    for m in re.finditer(r'const char \*(\w+)\s*=\s*"((?:[^"\\]|\\.|\\\n)*)";', text):
        out[m.group(1)] = m.group(2).replace("\\\n", "")
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in ("fasta.fa", "fastq.fq"):
        shutil.copyfile(os.path.join(REF, "test_data", f), os.path.join(OUT, f))
        os.chmod(os.path.join(OUT, f), 0o644)
    text = open(os.path.join(REF, "libms", "tests", "SA_test.cpp")).read()
    fasta_part, fastq_part = text.split("TEST(SATest, FastQTest)")
    a, b = literals(fasta_part), literals(fastq_part)
    fixture = {
        "source": "libms/tests/SA_test.cpp:11-136 (expected strings) + test_data/{fasta.fa,fastq.fq}",
        "FastaTest": {"file": "fasta.fa", "names": ["HSBGPG", "HSGLTH1"],
                      "sequences": [a["pExpectedSequenceFirst"], a["pExpectedSequenceSecond"]]},
        "FastQTest": {"illumina_file": "fasta.fa", "nanopore_file": "fastq.fq",
                      "illumina": [b["pExpectedISequenceFirst"], b["pExpectedISequenceSecond"]],
                      "nanopore": [b["pExpectedNSequenceFirst"], b["pExpectedNSequenceSecond"]]},
    }
    with open(os.path.join(OUT, "sa_test_expected.json"), "w") as f:
        json.dump(fixture, f, indent=1)
    print({k: [len(s) for s in v.get("sequences", v.get("nanopore", []))] for k, v in fixture.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
