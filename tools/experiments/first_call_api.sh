#!/bin/bash
# where the first msgpu_overlap_batched_ex call of a context spends its extra 25 ms: kernel + memory-copy timeline of the
# first and the second call of one context (tools/first_call_timing.py under rocprofv3)
OUT=gpurun_out/r3_05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/fctrace" -o t -- python3 tools/first_call_timing.py > "$OUT/first_call_traced.txt" 2>&1 || echo "trace failed"
tail -3 "$OUT/first_call_traced.txt"
ls "$OUT/fctrace"
python3 - "$OUT/fctrace" > "$OUT/first_call_timeline.txt" <<'PY'
import csv, sys, glob
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msgpu::", "")[:40]))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].endswith("k_index_pass1")]
# calls in the trace: warm (1), then per rep first + second; take rep 1: starts[-4] (first) and starts[-3] (second)
for label, k in (("FIRST call of a context", len(starts) - 4), ("SECOND call of the same context", len(starts) - 3)):
    lo = starts[k] - 3
    hi = starts[k + 1] - 3
    step = ev[lo:hi]
    t0 = prev_end = step[0][0]
    busy = 0
    print("==== %s" % label)
    for s, e, n in step:
        gap = s - prev_end
        if (e - s) > 20000 or gap > 20000:
            print("%9.1f us  dur %8.1f  gap %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n))
        busy += max(0, e - max(s, prev_end))
        prev_end = max(prev_end, e)
    print("wall %.1f us, busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
PY
cat "$OUT/first_call_timeline.txt" | head -120
rm -rf "$OUT/fctrace"
