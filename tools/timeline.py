#!/usr/bin/env python3
"""Timeline of the LAST call in a rocprofv3 trace: kernels (and memory copies if a
*_memory_copy_trace.csv sits beside the kernel trace), start and end relative to the first event
of that call (ms).  Usage: python tools/timeline.py <dir or kernel_trace.csv>"""
import csv
import glob
import os
import sys

arg = sys.argv[1]
files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*_kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"),
                     r["Kernel_Name"].split("(")[0].replace("curdle::", "").replace("void ", "")))
    mc = f.replace("_kernel_trace.csv", "_memory_copy_trace.csv")
    if os.path.exists(mc):
        for r in csv.DictReader(open(mc)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy")))
rows.sort()
gaps = [i for i in range(1, len(rows)) if rows[i][0] - max(r[1] for r in rows[max(0, i - 40):i]) > 2_000_000]
sel = rows[gaps[-1]:] if gaps else rows
t0 = sel[0][0]
for a, b, q, name in sel:
    print(f"{(a - t0) / 1e6:8.3f} {(b - t0) / 1e6:8.3f}  {q:>5}  {name[:60]}")
