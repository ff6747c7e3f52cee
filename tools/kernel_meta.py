#!/usr/bin/env python3
"""Extracts the gfx950 code object from libmcl_hip_engine.so (or another fat binary) and prints per-kernel register / LDS /
spill figures from its metadata; with --disasm NAME also the ISA of the kernels whose name contains NAME.

usage: tools/kernel_meta.py [lib.so] [--disasm k_rays_sweep] [--out /tmp/dev.co]"""
import re
import struct
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"


def extract(lib, out):
    data = open(lib, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle"
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, l = struct.unpack_from("<QQQ", data, off)
        off += 24
        name = data[off:off + l]
        off += l
        if b"gfx950" in name:
            open(out, "wb").write(data[i + o:i + o + s])
            return out
    raise SystemExit("no gfx950 code object")


def main():
    args = sys.argv[1:]
    lib = "monte_carlo_localization_amd/libmcl_hip_engine.so"
    out, dis = "/tmp/mcl_dev.co", None
    while args:
        a = args.pop(0)
        if a == "--disasm":
            dis = args.pop(0)
        elif a == "--out":
            out = args.pop(0)
        else:
            lib = a
    extract(lib, out)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for line in notes.splitlines():
        m = re.match(r"\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count):\s+(\S+)", line)
        if m:
            if m.group(1) == "name" and cur.get("name"):
                pass
            cur[m.group(1)] = m.group(2)
        if re.match(r"\s+\.wavefront_size", line) and cur.get("name"):
            rows.append(cur)
            cur = {}
    for r in rows:
        nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        nm = re.sub(r"\(.*", "", nm)
        print(f"{nm:45s} vgpr {r.get('vgpr_count','?'):>3} agpr {r.get('agpr_count','0'):>3} sgpr {r.get('sgpr_count','?'):>3} "
              f"vspill {r.get('vgpr_spill_count','0'):>3} sspill {r.get('sgpr_spill_count','0'):>3} lds {r.get('group_segment_fixed_size','0'):>6} "
              f"scratch {r.get('private_segment_fixed_size','0'):>5}")
    if dis:
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", out], capture_output=True, text=True).stdout
        on = False
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                on = dis in m.group(1)
            if on:
                print(line)


if __name__ == "__main__":
    main()
