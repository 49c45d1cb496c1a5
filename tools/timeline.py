#!/usr/bin/env python3
"""Timeline of the LAST call in a rocprofv3 kernel-trace CSV: kernel, stream/queue, start and end
relative to the first kernel of that call (ms).  Usage: python tools/timeline.py trace.csv [n_last_kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: kernels after the last k_synth / gap > 1 ms
gaps = [i for i in range(1, len(rows)) if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 300_000]
start = gaps[-1] if gaps else 0
sel = rows[start:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].split("(")[0].replace("curdle::", "")
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:8.3f} {(int(r["End_Timestamp"]) - t0) / 1e6:8.3f}  q{r.get("Queue_Id", "?"):>3}  {name[:60]}')
