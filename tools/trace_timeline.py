#!/usr/bin/env python3
"""One steady-state update of a rocprofv3 --kernel-trace csv as a timeline: every dispatch with its start offset,
duration and the idle gap before it, then the totals (busy, idle, span).  An update is the dispatches from one
k_resample_motion to the next.

usage: trace_timeline.py <kernel_trace.csv> [which update from the end, default 2; "first" = the first update of the run]
"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
marks = [i for i, r in enumerate(rows) if "k_resample_motion" in r[2]]
if len(sys.argv) > 2 and sys.argv[2] == "first":
    a, b = marks[0], marks[1]
else:
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    a, b = marks[-back - 1], marks[-back]
t0 = rows[a][0]
busy = idle = 0
prev_end = None
for s, e, name in rows[a:b]:
    gap = 0 if prev_end is None else s - prev_end
    name = name.split("(")[0]
    name = name if len(name) < 60 else name[:28] + ".." + name[-28:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
    busy += e - s
    idle += max(gap, 0)
    prev_end = max(e, prev_end or e)
span = rows[b][0] - t0
print(f"dispatches {b - a}  busy {busy / 1e3:.1f} us  idle {idle / 1e3:.1f} us  span to the next update {span / 1e3:.1f} us")
