#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace (and optional --pmc) CSV set into a small committed summary.

usage: summarize_rocprof.py <kernel_trace.csv> <out.md> [--pmc-fetch counter.csv] [--pmc-write counter.csv]
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    trace, out = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(trace)))
    by = defaultdict(list)
    for r in rows:
        by[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
    total = sum(d for v in by.values() for d, _ in v)
    lines = ["| kernel | grid | block | calls | avg us | min us | max us | % GPU time | VGPR | LDS B |", "|---|---|---|---|---|---|---|---|---|---|"]
    summ = {}
    for (k, g, b), v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        ds = [d for d, _ in v]
        r0 = v[0][1]
        lines.append(f"| {k} | {g} | {b} | {len(ds)} | {sum(ds)/len(ds)/1e3:.1f} | {min(ds)/1e3:.1f} | {max(ds)/1e3:.1f} | "
                     f"{100.0*sum(ds)/total:.2f} | {r0['VGPR_Count']} | {r0['LDS_Block_Size']} |")
        summ[f"{k}|grid={g}"] = {"calls": len(ds), "avg_us": sum(ds) / len(ds) / 1e3, "min_us": min(ds) / 1e3, "max_us": max(ds) / 1e3}
    extra = []
    args = sys.argv[3:]
    pmc = {}
    while args:
        flag, path = args[0], args[1]
        args = args[2:]
        vals = defaultdict(list)
        for r in csv.DictReader(open(path)):
            vals[(short(r["Kernel_Name"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, g, c), v in vals.items():
            pmc[f"{k}|grid={g}|{c}"] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)}
            extra.append(f"| {k} | {g} | {c} | {len(v)} | {sum(v)/len(v):.1f} | {max(v):.1f} |")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
        if extra:
            f.write("\n| kernel | grid | counter | launches | mean | max |\n|---|---|---|---|---|---|\n" + "\n".join(extra) + "\n")
    json.dump({"kernels": summ, "pmc": pmc}, open(out.rsplit(".", 1)[0] + ".json", "w"), indent=1)
    print("\n".join(lines[:8]))


main()
