#!/usr/bin/env python3
"""Writes profiles/<tag>_roofline_inputs.json, the static inputs of bench.py's `roofline` block, from committed evidence:

  * the dominant kernel's VALU instruction count and the clock it held -- rocprofv3 --pmc passes of the bench workload
    (counter_collection.csv: SQ_INSTS_VALU; GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the same dispatch);
  * the quad-cycles one wave64 VALU instruction of the kernel occupies a SIMD -- SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the same
    passes (1.0: four cycles whatever the kind) -- and the share of the wave-cycles the waves are parked in s_waitcnt / barriers
    (SQ_WAIT_ANY / SQ_WAVE_CYCLES);
  * memory-side traffic per launch -- FETCH_SIZE / WRITE_SIZE passes (KB; FETCH_SIZE doubled as the MI355X guide
    prescribes for gfx950, which counts 128-byte read requests at 64 bytes).

usage: roofline_inputs.py <tag> <sq.csv> <sq2.csv> <fetch.csv> <write.csv> <valu_rates.txt> <particles> <beams> [out dir, default profiles]
"""
import csv
import json
import re
import sys
from collections import defaultdict

tag, sq, sq2, fe, wr, ub, n, B = sys.argv[1:9]
outdir = sys.argv[9] if len(sys.argv) > 9 else "profiles"
KERNEL = "k_rays_sweep"


def counters(path):
    v = defaultdict(list)
    for r in csv.DictReader(open(path)):
        # (<true, ..> = the probe-counting build of bench.py's untimed update; <.., true> = the global-field form of long-range maps)
        # (the timed updates of the default workload run <false, false, REC, false>: LDS windows, one ray per lane)
        if KERNEL in r["Kernel_Name"] and "sweep<true" not in r["Kernel_Name"] and "sweep<false, true" not in r["Kernel_Name"]:
            v[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return v


def steady(rows):
    rows = rows[2:] if len(rows) > 3 else rows          # the first launches trace the spread cloud
    return sum(x for x, _ in rows) / len(rows), sum(d for _, d in rows) / len(rows)


a, b, f, w = counters(sq), counters(sq2), counters(fe), counters(wr)
insts, _ = steady(a["SQ_INSTS_VALU"])
lds, _ = steady(a["SQ_INSTS_LDS"])
gui, dur_ns = steady(b["GRBM_GUI_ACTIVE"])
clock_ghz = gui / 8.0 / dur_ns
fetch_kb, _ = steady(f["FETCH_SIZE"])
write_kb, _ = steady(w["WRITE_SIZE"])
active, _ = steady(b["SQ_ACTIVE_INST_VALU"]) if b.get("SQ_ACTIVE_INST_VALU") else (None, None)
wait_any, _ = steady(b["SQ_WAIT_ANY"]) if b.get("SQ_WAIT_ANY") else (None, None)
wave_cycles, _ = steady(a["SQ_WAVE_CYCLES"]) if a.get("SQ_WAVE_CYCLES") else (None, None)


def trip_mix_from_disassembly():
    """The probe trip of the beam walk as the shipped library contains it: opcode sequence of the loop body between the two
    `s_cbranch_execz` of MCL_SW_WALK, classified by the issue classes of profiles/*_op_rates.txt.  bench.py's VALU bound
    assumes 5 VALU instructions per trip; this records what the binary really has."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "monte_carlo_localization_amd", "libmcl_hip_engine.so")
    if not os.path.exists(lib):
        return None
    # (the instantiation the default workload runs: <COUNT = false, GLOBAL = false, REC = true, PAIRS = false, HYB = false>)
    txt = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_meta.py"), lib, "--disasm", "k_rays_sweepILb0ELb0ELb1ELb0ELb0E"],
                         capture_output=True, text=True).stdout
    ops = [re.sub(r"_e(32|64)$", "", l.split()[0]) for l in txt.splitlines() if l.startswith("\t")]
    want = ["v_mad_u64_u32", "v_mad_u64_u32", "v_mad_u32_u24", "ds_read_i8", "v_min3_u32", "s_waitcnt", "v_sub_co_u32", "s_andn2_b64"]
    hits = [i for i in range(len(ops) - len(want)) if ops[i:i + len(want)] == want]
    if not hits:
        return {"verified": False}
    valu = [o for o in want if o.startswith("v_")]
    return {"verified": True, "occurrences": len(hits), "valu_per_trip": len(valu), "sequence": want}


out = {
    "kernel": KERNEL, "particles": int(n), "beams": int(B),
    "valu_insts_per_launch": insts, "lds_insts_per_launch": lds, "simds": 1024, "clock_ghz": round(clock_ghz, 4),
    "quad_cycles_per_valu_inst": (active / insts) if active else None,
    "wait_any_frac": (wait_any / wave_cycles) if wait_any and wave_cycles else None,
    "kernel_ms_while_profiled": dur_ns / 1e6,
    "hbm_bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0,
    "hbm_bytes_source": f"profiles/{tag}_pmc_*.csv: 2 x FETCH_SIZE + WRITE_SIZE (KB) of {KERNEL}, steady-state launches; "
                        "8-byte accesses are outside the guide's calibration (16 B per lane), so this is an upper estimate",
    "fetch_kb": fetch_kb, "write_kb": write_kb,
    "probe_trip_in_binary": trip_mix_from_disassembly(),
    "sources": {"SQ_INSTS_VALU": f"profiles/{tag}_pmc_sq.csv", "GRBM_GUI_ACTIVE": f"profiles/{tag}_pmc_sq2.csv",
                "FETCH_SIZE": f"profiles/{tag}_pmc_fetch.csv", "WRITE_SIZE": f"profiles/{tag}_pmc_write.csv",
                "SQ_ACTIVE_INST_VALU, SQ_WAIT_ANY": f"profiles/{tag}_pmc_sq2.csv", "SQ_WAVE_CYCLES": f"profiles/{tag}_pmc_sq.csv"},
}
json.dump(out, open(f"{outdir}/{tag}_roofline_inputs.json", "w"), indent=1)
print(json.dumps(out, indent=1))
