#!/usr/bin/env python3
"""Wall time per update of the small and medium workloads: the reference's stock 2000-4000 particles x 61
beams, BASELINE configs[0] (4000 x 1081) and two sizes between the single-workgroup tail and k_rays_sweep.
graph_mode 0 = the default (three-launch path up to 8192 particles, hipGraph tail above), 1 = launch by launch.
usage: python tools/small_configs.py > profiles/<tag>_small_configs.txt   (on a GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth  # noqa: E402

m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
full = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
print("# Spielberg_map, init_particles_pose((0, 0, 0)), action (0.05, 0, 0.01), 300 updates, median of the last 250; ms per update")
for n, astep in ((2000, 18), (4000, 18), (4000, 1), (16384, 1), (65536, 18), (65536, 1), (262144, 18)):
    for gm in (0, 1):
        ang = synth.beam_angles()[::astep].copy()
        scan = full[::astep].copy()
        e = engine.Engine(max_particles=n, seed=42, graph_mode=gm)
        e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
        e.set_beam_angles(ang)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        ts = []
        for i in range(300):
            t0 = time.perf_counter()
            e.update((0.05, 0.0, 0.01), scan)
            ts.append(time.perf_counter() - t0)
        print("n=%-7d B=%-5d graph_mode=%d  %-13s median %.4f ms  min %.4f ms  pose %s" % (
            n, ang.size, gm, e.ray_kernel_name(), np.median(ts[50:]) * 1e3, np.min(ts) * 1e3, np.round(e.expected_pose(), 6)), flush=True)
        e.close()
