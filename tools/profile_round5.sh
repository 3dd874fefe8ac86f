#!/bin/bash
# round 5's evidence run: counters (tools/pmc_passes.sh), the default line, the kernel trace of the kernels-only command
out=${1:-gpurun_out/r5_prof}; mkdir -p "$out"
for f in muchsalsa_amd/csrc/*.hip muchsalsa_amd/csrc/*.cpp muchsalsa_amd/csrc/*.h include/*.h; do
  if [ "$f" -nt muchsalsa_amd/libmsgpu.so ]; then echo "STALE LIBRARY: $f"; exit 1; fi
done
bash tools/pmc_passes.sh "$out/pmc" cfg3 > "$out/pmc_passes.log" 2>&1; tail -2 "$out/pmc_passes.log"
cp "$out/pmc/pmc_summary.csv" "$out/pmc/pmc_meta.json" "$out/" 2>/dev/null
timeout -k 10 500 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || tail -20 "$out/bench_default.err"
python - "$out/bench_default.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.1f M/s  ms_per_step %.4f  roofline %.4f  traffic %s  valu %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("valu_issue_frac")))
print({k:v for k,v in d["config"].items() if k.startswith("survey_8d") and "note" not in k})
print("stages", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["stage_ms"].items() if k!="note"})
print("roofline_stages", {k:(round(v["frac"],3), v.get("traffic")) for k,v in d["roofline_stages"].items()})
for k in ("graph_stage","e2e","assemble_path","consensus","cpu_baseline","group"):
    v=d.get(k) or {}
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if isinstance(b,(int,float))})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/trace" -o kt -- python3 "$GRAFT_REPO_ROOT/bench.py" --kernels-only --steps 10 --warmup 2 > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.err"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
t=$(find "$out/trace" -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/timeline.py "$t" > "$out/timeline_one_step.txt" 2>/dev/null
rm -rf "$out/trace" "$out/pmc"/*.db 2>/dev/null
tail -3 "$out/timeline_one_step.txt"
