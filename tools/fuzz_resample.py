#!/usr/bin/env python3
"""Randomised resampling against the CPU oracle (run on a GPU box; test infrastructure, like tests/).

Every case draws a particle count around the thresholds of the engine's paths (single-workgroup tail, graph replay, full
CDF, compact parent list), a resampling mode, a seed and a weight pattern (uniform, smooth, a few heavy particles, mostly
zero, one survivor, weights that differ by 2^-40), sets the particles, runs updates and compares ALL resample indices of
every update with the oracle's (fixed-point weights of the weights given / of the log-weights the engine reports -> exact
CDF -> Philox draws; three updates per case, so the compact parent list of the second and third is exercised too).

usage: fuzz_resample.py [cases, default 40] [seed, default 1]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth      # noqa: E402
from oracle import oracle as orc                                  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc.build()
m = maps.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz"))
full = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncases):
    n = int(rng.choice([1000, 4000, 8192, 8193, 20000, 65536, 70000, 150000, 262144, 400000]))
    mode = int(rng.choice([engine.RESAMPLE_MULTINOMIAL, engine.RESAMPLE_SYSTEMATIC]))
    step = int(rng.choice([4, 9, 18]))
    ang = synth.beam_angles()[::step].copy()
    scan = full[::step].copy()
    eseed = int(rng.integers(1, 1 << 40))
    pattern = rng.choice(["uniform", "smooth", "heavy", "mostly-zero", "one", "tiny-steps"])
    w = np.full(n, 1.0)
    if pattern == "smooth":
        w = rng.uniform(0.0, 1.0, n)
    elif pattern == "heavy":
        w = rng.uniform(0.0, 1e-9, n)
        w[rng.choice(n, 5, replace=False)] = rng.uniform(0.5, 1.0, 5)
    elif pattern == "mostly-zero":
        w = np.zeros(n)
        k = max(1, n // 50)
        w[rng.choice(n, k, replace=False)] = rng.uniform(0.1, 1.0, k)
    elif pattern == "one":
        w = np.zeros(n)
        w[int(rng.integers(0, n))] = 1.0
    elif pattern == "tiny-steps":
        w = 1.0 + rng.integers(0, 4, n) * 2.0 ** -40
    w = w / w.sum()
    p = synth.tracking_cloud(np.random.default_rng(int(rng.integers(1, 1 << 30))), n)
    t0 = time.time()
    e = engine.Engine(max_particles=n, seed=eseed, resample_mode=mode)
    ok = True
    note = ""
    try:
        e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
        e.set_beam_angles(ang)
        e.set_particles(p, w)
        q_prev = orc.eng_quantize_weights(w)         # the fixed-point weights of the linear weights the engine was given
        for u in range(3):
            e.update((0.05, 0.0, 0.01), scan)
            idx = e.resample_indices()
            want = orc.eng_resample_indices(q_prev, 0, k53=orc.eng_philox_k53(eseed, u, 0, n)) if mode == engine.RESAMPLE_MULTINOMIAL \
                else orc.eng_resample_indices(q_prev, 1, k0=orc.eng_philox_k0(eseed, u))
            mism = int(np.count_nonzero(idx != want))
            if mism:
                ok = False
                note += f" update {u}: {mism} indices differ"
            _, q_prev, _ = orc.eng_weights_from_log(e.log_weights())
    finally:
        e.close()
    bad += not ok
    print(f"case {case:3d} n {n:6d} mode {mode} beams {ang.size:4d} {pattern:11s} {'OK' if ok else 'MISMATCH' + note} {time.time() - t0:.1f}s", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
