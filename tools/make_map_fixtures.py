#!/usr/bin/env python3
"""Turns the reference's map DATA files (maps/*.yaml + image) into compact occupancy-grid
fixtures under tests/golden/ (bit-packed occupied/unknown masks).  Runs only in the build
container, where /root/reference exists; the GPU box sees only the committed .npz files."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from monte_carlo_localization_amd import maps  # noqa: E402

REF = "/root/reference/maps"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
for name in ("Spielberg_map", "sibal1", "icra_2_clean", "first_map"):
    m = maps.load_map_yaml(os.path.join(REF, name + ".yaml"))
    occ = int((m.data > 50).sum()); fr = int((m.data == 0).sum()); unk = int((m.data < 0).sum())
    print(f"{name}: {m.width}x{m.height} res={float(m.resolution)!r} origin=({m.origin_x},{m.origin_y}) occ/free/unk={occ}/{fr}/{unk}")
    maps.save_npz(m, os.path.join(OUT, f"map_{name}.npz"))
