#!/bin/bash
# One capture of a round's state into gpurun_out/<name>/: default bench line, kernel-trace stats of the same command,
# counter passes.  Copy the summaries (bench_default.json, kernel_stats.csv, pmc_summary.csv, pmc_meta.json) to
# profiles/<name>/ afterwards.   usage: tools/profile_round.sh NAME
set -u
NAME=$1; OUT=gpurun_out/$NAME
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || echo "bench failed"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 bench.py --steps 10 --warmup 2 --kernels-only > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || echo "trace failed"
cp "$OUT"/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
KT=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python3 tools/timeline.py "$KT" > "$OUT/timeline_one_step.txt"
# the same with every leg of the default line that launches kernels (host-to-host, contraction, consensus gather,
# assemblePath: gather + FASTA wrapping + edit-distance meter, the tiled-unitig leg): kernel_stats_all_legs.csv
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_all" -o trace -- python3 bench.py --steps 10 --warmup 2 --cpu-sample-reads 0 --no-e2e > "$OUT/bench_all_legs_under_rocprof.json" 2> "$OUT/bench_all_legs_under_rocprof.err" || echo "all-legs trace failed"
find "$OUT/trace_all" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_all_legs.csv" \;
rm -rf "$OUT/trace_all"
bash tools/pmc_passes.sh "$OUT/pmc" cfg3
cp "$OUT/pmc/pmc_summary.csv" "$OUT/pmc/pmc_meta.json" "$OUT/" 2>/dev/null
rm -rf "$OUT/trace"/*.db
ls -la "$OUT"
