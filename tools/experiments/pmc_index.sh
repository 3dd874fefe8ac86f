#!/bin/bash
# counters of the index-build kernels (two SQ passes): gpurun_out/pmc_index/
OUT=gpurun_out/pmc_index; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -o "pmc_$name" -- python3 bench.py --steps 2 --warmup 1 --kernels-only > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM
python3 tools/pmc_summarize.py "$OUT" cfg3 1 > /dev/null
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/pmc_index/pmc_summary.csv")))
for r in rows:
    if "index" in r["kernel"] or "k_sort" in r["kernel"]:
        w=float(r["SQ_WAVES"])
        print(r["kernel"], "waves %d" % w, " per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f SMEM %.1f | wave quad-cycles %.0f wait_any %.0f wait_inst %.0f active %.0f | lds conflict %.0f of %.0f" % (
            float(r["SQ_INSTS_VALU"])/w, float(r["SQ_INSTS_SALU"])/w, float(r["SQ_INSTS_LDS"])/w, float(r["SQ_INSTS_VMEM_RD"])/w, float(r["SQ_INSTS_VMEM_WR"])/w, float(r["SQ_INSTS_SMEM"])/w,
            float(r["SQ_WAVE_CYCLES"])/w, float(r["SQ_WAIT_ANY"])/w, float(r["SQ_WAIT_INST_ANY"])/w, float(r["SQ_ACTIVE_INST_ANY"])/w, float(r["SQ_LDS_BANK_CONFLICT"])/w, float(r["SQ_LDS_IDX_ACTIVE"])/w))
PY
