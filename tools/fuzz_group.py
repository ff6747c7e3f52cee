#!/usr/bin/env python3
"""Randomised sharded sets against one engine (run on a GPU box; test infrastructure, like tests/).

Every case draws a shard count (2-5, all on device 0), a shard size around the path thresholds, a resampling mode, a map, a
beam step, how the set starts (pose cloud, uniform over the free cells, given particles with skewed weights) and runs a few
updates through mcl_group_* and through ONE engine that holds all particles: resample indices and particles must be
identical, weights and pose equal to the last bits of a sum taken in a different order.

usage: fuzz_group.py [cases, default 30] [seed, default 1]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth      # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MAPS = {"spielberg": maps.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz")), "sibal1": maps.load_npz(os.path.join(GOLDEN, "map_sibal1.npz"))}
full = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncases):
    shards = int(rng.integers(2, 6))
    n_per = int(rng.choice([1024, 4096, 8192, 16384, 40000, 70000, 131072]))
    n = n_per * shards
    mode = int(rng.choice([engine.RESAMPLE_MULTINOMIAL, engine.RESAMPLE_SYSTEMATIC]))
    mname = rng.choice(list(MAPS))
    m = MAPS[mname]
    step = int(rng.choice([4, 9, 18]))
    ang = synth.beam_angles()[::step].copy()
    scan = full[::step].copy()
    eseed = int(rng.integers(1, 1 << 40))
    start = rng.choice(["pose", "global", "given"])
    t0 = time.time()
    one = engine.Engine(max_particles=n, seed=eseed, resample_mode=mode)
    grp = engine.Group([0] * shards, max_particles=n_per, seed=eseed, resample_mode=mode)
    note = ""
    try:
        for x in (one, grp):
            x.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
            x.set_beam_angles(ang)
        if start == "pose":
            one.init_particles_pose((0.0, 0.0, 0.0), n); grp.init_particles_pose((0.0, 0.0, 0.0), n)
        elif start == "global":
            one.init_global(n); grp.init_global(n)
        else:
            p = synth.tracking_cloud(np.random.default_rng(int(rng.integers(1, 1 << 30))), n)
            w = rng.uniform(0.0, 1.0, n) ** 8
            w[rng.choice(n, n // 3, replace=False)] = 0.0
            if rng.random() < 0.3:
                w[: n_per] = 0.0                      # a shard that carries no weight
            w /= w.sum()
            one.set_particles(p, w); grp.set_particles(p, w)
        if not np.array_equal(grp.get_particles(), one.get_particles()):
            note += " start differs"
        for u in range(4):
            one.update((0.05, 0.0, 0.01), scan); grp.update((0.05, 0.0, 0.01), scan)
            if not np.array_equal(grp.resample_indices(), one.resample_indices()):
                note += f" update {u}: indices"
            if not np.array_equal(grp.get_particles(), one.get_particles()):
                note += f" update {u}: particles"
            if not np.allclose(grp.get_weights(), one.get_weights(), rtol=1e-12, atol=0):
                note += f" update {u}: weights"
            if not np.allclose(grp.expected_pose(), one.expected_pose(), rtol=0, atol=1e-11):
                note += f" update {u}: pose"
        xb = grp.exchange_bytes()
    finally:
        grp.close(); one.close()
    bad += bool(note)
    print(f"case {case:3d} {mname:9s} shards {shards} x {n_per:6d} mode {mode} beams {ang.size:4d} {start:6s} lists {int(bool(xb['lists']))} "
          f"{'OK' if not note else 'MISMATCH' + note} {time.time() - t0:.1f}s", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
