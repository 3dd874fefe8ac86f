#!/bin/bash
# kernel timeline of one weak-scaling step at world 1 over RCCL (bench.py --force-dist): where the exchange's 0.3 ms go
OUT=${1:-gpurun_out/forcedist_tl}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -o trace -- python3 bench.py --steps 10 --warmup 2 --kernels-only --force-dist > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "trace failed"
KT=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$KT" 6 > $OUT/timeline_step.txt
rm -rf "$OUT/trace"
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['stage_ms'])"
cat $OUT/timeline_step.txt
