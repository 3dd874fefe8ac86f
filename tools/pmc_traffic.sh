#!/bin/bash
OUT=$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT" -o "pmc_$c" -- python3 bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --no-consensus > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err" || echo "pass $c failed"
done
