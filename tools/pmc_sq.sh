#!/bin/bash
OUT=$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -o "pmc_$name" -- python3 bench.py --steps 3 --warmup 1 --kernels-only --workload cfg3 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH
python3 tools/pmc_summarize.py "$OUT" cfg3 1
