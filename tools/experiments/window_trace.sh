#!/bin/bash
# kernel timeline of ONE lean dispatcher job (rows in pinned host memory -> tables in pinned host memory): what the windows
# spend their time on.   tools/experiments/window_trace.sh <out dir>
out=${1:-gpurun_out/window_trace}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -o trace -- python3 tools/experiments/lean_wall.py traced 0 > $out/run.txt 2> $out/run.err || { echo "trace failed"; tail -5 $out/run.err; exit 1; }
cat $out/run.txt
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
kt = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msgpu::", "")[:40], "q" + r.get("Queue_Id", "?")))
for f in mc:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")), "dma"))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].endswith("k_index_bin")]
i0 = starts[-1]
# the copy of the rows in front of it
j = i0
while j > 0 and not ev[j][2].startswith("COPY"): j -= 1
t0 = ev[j][0]
with open(out + "/timeline.txt", "w") as f:
    for s, e, n, q in ev[j:]:
        if e - s < 3000 and not n.startswith("COPY"): continue   # kernels of 3 us and more
        f.write("%9.1f %8.1f  %s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, n, q))
print(open(out + "/timeline.txt").read())
PY
rm -rf $out/trace
