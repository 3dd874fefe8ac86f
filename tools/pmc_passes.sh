#!/bin/bash
# rocprofv3 counter passes for bench.py (one --pmc set per run; never combined with trace domains other than kernel-trace),
# then per-kernel averages -> OUTDIR/pmc_summary.csv + pmc_meta.json (tools/pmc_summarize.py).
# usage: tools/pmc_passes.sh OUTDIR [workload]
set -u
OUT=$1; WL=${2:-cfg3}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -o "pmc_$name" -- python3 bench.py --steps 3 --warmup 1 --kernels-only --workload "$WL" > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
# the gather kernel of the consensus leg (not part of --kernels-only): its two traffic counters
rung() { # name, counter
  timeout -k 10 300 rocprofv3 --pmc "$2" --output-format csv -d "$OUT" -o "pmc_$1" -- python3 bench.py --steps 3 --warmup 1 --cpu-sample-reads 0 --assemble-window-mb -1 --batches 0 --no-tiled --no-e2e --workload "$WL" > "$OUT/pmc_$1.json" 2> "$OUT/pmc_$1.err" || echo "pass $1 failed"
}
rung zfetch_gather FETCH_SIZE
rung zwrite_gather WRITE_SIZE
python3 tools/pmc_summarize.py "$OUT" "$WL" 1
ls "$OUT"
