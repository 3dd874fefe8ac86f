#!/bin/bash
# One development cycle on the GPU box: parity tests (fresh device memory poisoned), the full-size and dispatcher tests, the default
# workload's kernels-only line, and a kernel trace of it.   tools/gpu_cycle.sh OUTDIR [quick]
out=${1:-gpurun_out/cycle}; mkdir -p "$out"
set -o pipefail
# the library that travelled must be newer than every kernel source (a failed local build leaves the old one behind)
for f in muchsalsa_amd/csrc/*.hip muchsalsa_amd/csrc/*.cpp muchsalsa_amd/csrc/*.h include/*.h; do
  if [ "$f" -nt muchsalsa_amd/libmsgpu.so ]; then echo "STALE LIBRARY: $f is newer than libmsgpu.so"; exit 1; fi
done
MSGPU_POISON=1 timeout -k 10 200 python -m pytest --timeout 60 tests/test_gpu_parity.py -x -q -k "not bench and not rehearsal and not exchange" > "$out/pytest_parity_poison.log" 2>&1 || { tail -30 "$out/pytest_parity_poison.log"; echo "PARITY FAILED"; exit 1; }
tail -2 "$out/pytest_parity_poison.log"
if [ "$2" != "quick" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_batched.py tests/test_golden_fixtures.py -x -q -m gpu > "$out/pytest_full.log" 2>&1 || { tail -30 "$out/pytest_full.log"; echo "FULLSIZE FAILED"; exit 1; }
  tail -2 "$out/pytest_full.log"
fi
timeout -k 10 300 python bench.py --kernels-only --steps 20 --warmup 3 > "$out/bench_kernels_only.json" 2> "$out/bench_kernels_only.err" || { tail -20 "$out/bench_kernels_only.err"; echo "BENCH FAILED"; exit 1; }
python - "$out/bench_kernels_only.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("ms_per_step %.4f  value %.1f M/s  stages %s  roofline %.4f" % (d["ms_per_step"], d["value"]/1e6, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["stage_ms"].items() if k!="note"}, d["roofline"]["frac"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/trace" -o kt -- python3 "$GRAFT_REPO_ROOT/bench.py" --kernels-only --steps 10 --warmup 2 > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.err" || { tail -20 "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.err"; echo "ROCPROF FAILED"; exit 1; }
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv" && head -40 "$out/kernel_stats.csv"
t=$(find "$out/trace" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/timeline.py "$t" > "$out/timeline_one_step.txt" 2>/dev/null && tail -45 "$out/timeline_one_step.txt"
rm -rf "$out/trace"
