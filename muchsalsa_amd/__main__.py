"""python -m muchsalsa_amd <contigs.paf> <unitigs.fa> <nanopore.fa|fq> <outdir> [threads] [wiggleRoom=300]
(the argument list of the reference executable, src/Application.cpp:34-39)"""
import json
import sys

from . import _lib

_lib.PRELOAD_TORCH = False  # this process never imports torch: load libmsgpu against the system HIP runtime directly

from .pipeline import run  # noqa: E402


def main(argv):
    if len(argv) < 4:
        sys.stderr.write(__doc__ + "\n")
        return -1
    threads = int(argv[4]) if len(argv) > 4 else None
    wiggle = int(argv[5]) if len(argv) > 5 else 300
    timings = {}
    out = run(argv[0], argv[1], argv[2], argv[3], threads, wiggle, timings=timings)
    out["seconds"] = {k: round(v, 4) for k, v in timings.items()}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
