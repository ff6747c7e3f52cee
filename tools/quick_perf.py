#!/usr/bin/env python3
"""Ad-hoc timing of the update at a given size (development aid; bench.py is the contract)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    kern = sys.argv[3] if len(sys.argv) > 3 else "skip"
    regime = sys.argv[4] if len(sys.argv) > 4 else "tracking"
    rpl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    astep = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
    ang = synth.beam_angles()[::astep].copy()
    e = engine.Engine(max_particles=n, seed=42, ray_kernel={"march": engine.RAYS_MARCH, "skip": engine.RAYS_SKIP, "quad": engine.RAYS_QUAD, "cell": engine.RAYS_CELL}.get(kern, engine.RAYS_AUTO), rays_per_lane=rpl)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(ang)
    scan = synth.scan_from_pose(e, m, ang, (0.0, 0.0, 0.0))
    rng = np.random.default_rng(42)
    p = synth.tracking_cloud(rng, n) if regime == "tracking" else synth.global_cloud(rng, m, n)
    e.set_particles(p, np.full(n, 1.0 / n))
    for i in range(steps):
        t0 = time.perf_counter()
        e.update((0.05, 0.0, 0.01), scan)
        dt = time.perf_counter() - t0
        print(f"update {i}: wall {dt*1e3:.2f} ms  stages {np.round(e.stage_timings(),3)} ray_kernel {e.ray_kernel_ms():.3f}  counters {e.counters()}  pose {e.expected_pose()}", flush=True)
    print("particle*beam/s (last):", n * ang.size / dt)

main()
