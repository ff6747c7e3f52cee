#!/usr/bin/env python3
"""Long runs of the round-5 forms of the ray kernel against each other (run on a GPU box; test infrastructure, like tests/): the same
seeded filter through (a) the default choice (k_rays_sweep turning the beam direction, one ray per lane in LDS windows), (b)
MCL_SWEEP_NO_REC=1 (directions fetched per ray), (c) MCL_SWEEP_PAIRS=1 (two rays per lane), (d) MCL_SWEEP_GLOBAL=1 (the wedge
fields probed in global memory, pairs; MCL_SWEEP_HYBRID=0), (d') the hybrid form (MCL_SWEEP_GLOBAL=1 MCL_SWEEP_HYBRID=2), (e) MCL_SWEEP_GLOBAL=1 + MCL_SWEEP_NO_REC=1, (f) MCL_NO_STALE_LAYOUT=1 -- particles, weights
and resample indices after N updates must be bit-identical (sha256), on the counting-sort sizes and on the radix-sort size,
tracking and global regime, both resampling modes; then 300 updates at 4M x 1081 with an oracle spot check of the last update's
log-weights.

usage: soak_round5.py [updates, default 200]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth      # noqa: E402
from oracle import oracle as orc                                  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
full = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
VARIANTS = {"default": {}, "fetched-directions": {"MCL_SWEEP_NO_REC": "1"}, "pairs": {"MCL_SWEEP_PAIRS": "1"},
            "global-fields": {"MCL_SWEEP_GLOBAL": "1", "MCL_SWEEP_HYBRID": "0"}, "global-fields-fetched": {"MCL_SWEEP_GLOBAL": "1", "MCL_SWEEP_NO_REC": "1"},
            "hybrid": {"MCL_SWEEP_GLOBAL": "1", "MCL_SWEEP_HYBRID": "2"},        # LDS windows + the global fields where a ray leaves its window
            "own-layout": {"MCL_NO_STALE_LAYOUT": "1"}}


def run(n, astep, nsteps, regime, mode, env):
    for k, v in env.items():
        os.environ[k] = v
    ang = synth.beam_angles()[::astep].copy()
    scan = full[::astep].copy()
    e = engine.Engine(max_particles=n, seed=7, resample_mode=mode)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(ang)
    for k in env:                    # (read at mcl_create / mcl_set_map / mcl_set_beam_angles)
        os.environ.pop(k, None)
    if regime == "tracking":
        e.init_particles_pose((0.0, 0.0, 0.0), n)
    else:
        e.init_global(n)
    rng = np.random.default_rng(5)
    t0 = time.perf_counter()
    obs = scan
    for _ in range(nsteps):
        obs = np.clip(scan + rng.normal(0, 0.02, scan.size), 0, 30).astype(np.float32)
        e.update((0.05, 0.0, 0.01), obs)
    dt = (time.perf_counter() - t0) / nsteps * 1e3
    h = hashlib.sha256(e.get_particles().tobytes() + e.get_weights().tobytes() + e.resample_indices().tobytes()).hexdigest()[:16]
    return h, round(dt, 3), e.ray_kernel_name(), e, ang, obs


bad = 0
for n, astep in ((262144, 2), (1048576, 2), (4194304, 4)):
    for regime in ("tracking", "global"):
        for mode in (0, 1):
            if n == 4194304 and (regime, mode) != ("tracking", 0):
                continue
            ref = None
            for name, env in VARIANTS.items():
                h, ms, kern, e, _, _ = run(n, astep, steps, regime, mode, env)
                e.close()
                ref = ref or h
                same = h == ref
                bad += not same
                print(f"n {n:8d} beams {1081 // astep + (1081 % astep > 0):4d} {regime:8s} mode {mode} {name:20s} {kern:12s} {ms:7.3f} ms/update  sha {h}  "
                      f"{'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
h, ms, kern, e, ang, obs = run(4 << 20, 1, 300, "tracking", 0, {})
om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
parts, lw = e.get_particles(), e.log_weights()
pick = np.random.default_rng(1).choice(4 << 20, 4000, replace=False)
L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
logw, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts[:, pick]), ang, orc.obs_index(obs, om), L)
ok = np.array_equal(lw[pick], logw)
bad += not ok
print(f"4M x 1081, 300 updates: {ms} ms/update, sha {h}, 4000 sampled log-weights of the last update vs the oracle: {'EQUAL' if ok else 'MISMATCH'}; counters {e.counters()}")
e.close()
print("differences:", bad)
sys.exit(1 if bad else 0)
