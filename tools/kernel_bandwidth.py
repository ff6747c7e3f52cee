#!/usr/bin/env python3
"""Per-kernel achieved bandwidth of the streaming kernels of one update, from a committed rocprofv3 kernel trace summary
(tools/summarize_rocprof.py's JSON) and each kernel's ALGORITHMIC bytes per particle -- what the kernel must read and write
once, as its source states it, not what the caches moved ("the streaming kernels K1/K2/K4-K7 are individually
HBM-bound and should each reach >= 50 % of HBM peak").

usage: tools/kernel_bandwidth.py <kernel_trace_summary.json> <particles> <out.md> [pmc_sq.csv]

With the SQ counter pass (profiles/rNN_pmc_sq.csv: SQ_INSTS_VALU per launch) a last column prices each kernel's VALU instructions
at 4 issue cycles each on 1024 SIMDs x 2.4 GHz (fp64 and most integer instructions are 4-cycle, profiles/r02_op_rates.txt): the
time the instruction stream alone needs -- what a kernel far below the HBM figure is waiting for, measured, not guessed.

min_us (the steady-state launches; the first launches of a bench run work on the spread cloud) is the duration used.
"""
import json
import sys

HBM_PEAK_GBS = 8000.0

# kernel -> (bytes read, bytes written) per particle, what they are, what bounds the kernel when it is not HBM
ALG = {
    "k_resample_motion": (32 + 8, 24 + 4 + 32 + 8 + 4,
                          "R: the parent's 32-byte record out of the compact list + its search (ctop / ccdf, cache-resident); "
                          "W: x, y, theta, parent index, the ray stage's constants (cos, sin, px, py), zeroed accumulator and far flags",
                          "VALU: two Philox4x32-10 calls, fp64 log + sqrt twice, four fp64 sincos (Box-Muller, motion, heading "
                          "constants) per child; the gather hits a list of a few MB"),
    "k_sort_keys": (32 + 8, 4 + 4, "R: constants + heading; W: 22-bit key + index", "at HBM / Infinity-Cache speed"),
    "k_sort_gather": (4 + 32 + 8 + 8, 32 + 8 + 4,
                      "R: sorted index, then the constants and the heading of THAT particle (two scattered fetches), neighbouring "
                      "keys for the unit cuts; W: sorted constants, headings, slot -> particle",
                      "scattered 32-byte + 8-byte reads: a 128-byte line is fetched for each (over-fetch x3-x4)"),
    "k_unit_sums": (32, 0, "R: sorted constants (bounding box per unit)", ""),
    "k_combine_logw": (8 + 4, 8, "R: slot accumulator, slot -> particle; W: log-weight at the particle's index",
                       "one scattered 8-byte store per particle (a partial line write each)"),
    "k_weights": (8 + 24, 8 + 8, "R: log-weight, x, y, theta; W: weight, fixed-point weight", "fp64 exp + sincos per particle beside the stream"),
    "k_scan_partials": (8, 0, "R: fixed-point weights", "small at this size: launch-bound below ~10 us"),
    "k_scan_final": (8, 8 + 0.5, "R: fixed-point weights; W: CDF, group leaders (+ 44 B per WEIGHTED particle: the compact list)", ""),
    "k_cell_bbox": (32 / 16.0, 0, "R: every 16th particle's constants", "launch-bound"),
}
RADIX = ("radix_sort_onesweep", 3 * 16 + 8, "rocPRIM onesweep, 22-bit keys + 32-bit values: 3 digit passes of (8 B read + 8 B written) + the histogram pass")


def main():
    summ, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    ks = json.load(open(summ))["kernels"]
    valu = {}
    if len(sys.argv) > 4:
        import csv
        from collections import defaultdict
        acc = defaultdict(list)
        for r in csv.DictReader(open(sys.argv[4])):
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcl::", "")].append(float(r["Counter_Value"]))
        valu = {k: min(v) for k, v in acc.items()}
    rows = []
    for name, (rd, wr, what, bound) in ALG.items():
        hits = [(k, v) for k, v in ks.items() if k.split("|")[0].replace("mcl::", "") == name]
        if not hits:
            continue
        k, v = max(hits, key=lambda kv: kv[1]["calls"])
        us = v["min_us"]
        gbs = (rd + wr) * n / (us * 1e-6) / 1e9
        iv = valu.get(name)
        rows.append((name, rd, wr, us, gbs, gbs / HBM_PEAK_GBS, what, bound, iv))
    # the library sort: all its launches of one update together
    rad = [(k, v) for k, v in ks.items() if RADIX[0] in k]
    upd = max((v["calls"] for k, v in ks.items() if "k_sort_gather" in k), default=0)
    if rad and upd:
        us = sum(v["avg_us"] * v["calls"] for _, v in rad) / upd          # all its launches of one update
        gbs = RADIX[1] * n / (us * 1e-6) / 1e9
        rows.append(("rocprim radix_sort_pairs (%.0f launches per update)" % (sum(v["calls"] for _, v in rad) / upd), RADIX[1] / 2.0, RADIX[1] / 2.0, us,
                     gbs, gbs / HBM_PEAK_GBS, RADIX[2], "a library sort (rocPRIM onesweep), its small launches included", None))
    lines = [f"Streaming kernels of one update at {n} particles: algorithmic bytes per particle / steady-state duration (min over the traced "
             f"launches) against {HBM_PEAK_GBS / 1000:.0f} TB/s.  Source: {summ}", "",
             "| kernel | B read | B written | us | GB/s | of HBM peak | VALU insts / launch | their issue time, us | bytes are | below 50 % because |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for name, rd, wr, us, gbs, frac, what, bound, iv in rows:
        issue = f"{iv * 4.0 / (1024 * 2.4e9) * 1e6:.1f}" if iv else ""
        lines.append(f"| {name} | {rd:g} | {wr:g} | {us:.1f} | {gbs:.0f} | {100 * frac:.0f} % | {iv:.3g} | {issue} | {what} | {bound if frac < 0.5 else ''} |"
                     if iv else f"| {name} | {rd:g} | {wr:g} | {us:.1f} | {gbs:.0f} | {100 * frac:.0f} % | | | {what} | {bound if frac < 0.5 else ''} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
