#!/usr/bin/env python3
"""Timeline of one timed step from a rocprofv3 kernel trace: start, duration and the idle gap before every dispatch.
python tools/timeline.py <trace_kernel_trace.csv> [step index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msgpu::", "")[:44], r.get("Queue_Id")) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2].endswith(("k_index_pass1", "k_index_bin"))]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
step = ev[starts[k] - 1:starts[k + 1] - 1]
t0, prev_end, idle = step[0][0], step[0][0], 0
for s, e, n, q in step:
    gap = s - prev_end
    if gap > 0:
        idle += gap
    print("%8.1f %8.1f  gap %6.1f  %s q%s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n, q))
    prev_end = max(prev_end, e)
print("wall %.1f us, idle %.1f us" % ((prev_end - t0) / 1e3, idle / 1e3))
