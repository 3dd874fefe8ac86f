#!/bin/bash
# per-kernel times of the timed steps: tools/ktrace.sh NAME [bench args] -> gpurun_out/NAME_kernel_stats.csv (+ a short table on stdout)
NAME=$1; shift
OUT=gpurun_out/$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 bench.py --steps 10 --warmup 2 --kernels-only "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "trace failed"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/${NAME}_kernel_stats.csv" \;
rm -rf "$OUT/trace"
python3 - "gpurun_out/${NAME}_kernel_stats.csv" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
# steps in the trace = launches of the index pass (warm-up + timed + the stage-marker steps bench.py adds)
steps=max([int(r["Calls"]) for r in rows if "k_compact" in r["Name"]] or [1])
for r in rows:
    n=re.sub(r"\(.*","",r["Name"]).replace("void ","").replace("msgpu::","")
    if "rocclr" in n or "at::" in n: continue
    calls=int(r["Calls"]); avg=float(r["AverageNs"])/1e3
    per_step=float(r["TotalDurationNs"])/1e3/steps
    tot+=per_step
    print("%-34s calls %4d  avg %9.1f us  per step %8.1f us" % (n[:34],calls,avg,per_step))
print("sum per step %.1f us (%d steps in the trace)" % (tot, steps))
PY
python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['stage_ms'])"
