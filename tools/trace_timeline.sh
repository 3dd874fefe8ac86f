#!/bin/bash
# kernel trace of the timed steps + the timeline of one step: tools/trace_timeline.sh NAME [bench args]
#   -> gpurun_out/NAME_kernel_stats.csv, gpurun_out/NAME_timeline.txt
NAME=$1; shift
OUT=gpurun_out/$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 bench.py --steps 10 --warmup 2 --kernels-only "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "trace failed"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/${NAME}_kernel_stats.csv" \;
KT=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$KT" > "gpurun_out/${NAME}_timeline.txt"
rm -rf "$OUT/trace"
python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['stage_ms'])"
