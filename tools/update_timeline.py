#!/usr/bin/env python3
"""One steady-state update as a timeline: every kernel of the LAST complete update in a rocprofv3 kernel trace (CSV), with its
start relative to the update's first kernel, its duration and the idle gap before it (no kernel of the process running).
usage: tools/update_timeline.py <kernel_trace.csv> [first-kernel substring, default k_resample_motion] [out.md]"""
import csv
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_resample_motion"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcl::", "")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    a, b = starts[-3], starts[-2]                      # the last-but-one complete update (the last may be the parity step)
    # an update begins with whatever precedes its first kernel since the previous update's last kernel: take [a, b)
    seg = rows[a:b]
    t0 = seg[0][0]
    lines = ["| kernel | start, us | duration, us | idle before, us |", "|---|---|---|---|"]
    busy_until, idle = t0, 0.0
    for s, e, n in seg:
        gap = max(0, s - busy_until) / 1e3
        idle += gap
        lines.append(f"| {n[:60]} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} |")
        busy_until = max(busy_until, e)
    total = (rows[b][0] - t0) / 1e3
    lines.append("")
    lines.append(f"update: {total:.1f} us from its first kernel to the next update's first kernel; {len(seg)} kernels; "
                 f"idle between kernels {idle:.1f} us + {(rows[b][0] - busy_until) / 1e3:.1f} us before the next update starts")
    out = "\n".join(lines)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
