#!/usr/bin/env python3
"""Randomised configurations of the ray stage against the CPU oracle (run on a GPU box; test infrastructure, like tests/).

Every case draws a map, a maximum range, a beam set (count, span up to a full turn, even or jittered spacing), a particle
count around the thresholds of the kernels' paths and a cloud (tracking, uniform over the free cells, a few far-apart
clusters, a mixture with stragglers and non-finite rows), lets the engine choose its kernels (RAYS_AUTO; a quarter of the
cases with MCL_SWEEP_GLOBAL=1, ranges of up to 30 m = 600 px take the global-field form anyway), and compares the
log-weights of sampled particles after sensor_update and after one full update with the oracle's, bit for bit.

usage: fuzz_ray_stage.py [cases, default 40] [seed, default 1]
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
MAPS = {
    "spielberg": maps.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz")),
    "sibal1": maps.load_npz(os.path.join(GOLDEN, "map_sibal1.npz")),
    "levine": maps.synthetic_levine(),
}
full = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncases):
    mname = rng.choice(list(MAPS))
    m = MAPS[mname]
    # (14 - 30 m: 241 .. 600 px on these maps -- beyond the 256-cell LDS windows, k_rays_sweep's global-field form)
    max_range = float(rng.choice([12.0, 12.0, 8.0, 5.0, 14.0, 20.0, 30.0]))
    os.environ["MCL_SWEEP_GLOBAL"] = "1" if rng.random() < 0.25 else "0"          # read at mcl_create: the global-field form on any range
    B = int(rng.choice([61, 181, 271, 361, 541, 721, 1000, 1081, 1440]))
    span = float(rng.choice([1.5 * np.pi, 1.5 * np.pi, np.pi, 0.5 * np.pi, 1.9 * np.pi, 2.0 * np.pi * (B - 1) / B]))
    a0 = float(rng.choice([-0.5 * span, -0.5 * span, -0.75 * np.pi, 0.3]))
    ang = a0 + np.arange(B) * (span / (B - 1))
    spacing = rng.choice(["even", "even", "jitter"])
    if spacing == "jitter":
        ang = np.sort(ang + rng.uniform(-0.4, 0.4, B) * (span / (B - 1)))
    ang = ang.astype(np.float32)
    if not np.all(np.diff(ang) > 0):
        continue
    n = int(rng.choice([20000, 66000, 100000, 150000, 262144, 300000]))
    kind = rng.choice(["tracking", "uniform", "clusters", "mixture"])
    free = np.argwhere(np.asarray(m.data).reshape(m.height, m.width) == 0)
    res = float(np.float32(m.resolution))

    def at_free(k):
        c = free[rng.integers(0, len(free), k)]
        return np.stack([m.origin_x + (c[:, 1] + rng.uniform(0.1, 0.9, k)) * res, m.origin_y + (c[:, 0] + rng.uniform(0.1, 0.9, k)) * res,
                         rng.uniform(-np.pi, np.pi, k)])
    if kind == "tracking":
        c0 = at_free(1)[:, 0]
        p = c0[:, None] + rng.normal(0, 1, (3, n)) * np.array([[0.5], [0.5], [0.4]])
    elif kind == "uniform":
        p = at_free(n)
    elif kind == "clusters":
        cs = at_free(int(rng.integers(2, 12)))
        p = cs[:, rng.integers(0, cs.shape[1], n)] + rng.normal(0, 1, (3, n)) * np.array([[0.15], [0.15], [0.2]])
    else:
        p = at_free(n)
        k = n // 2
        c0 = at_free(1)[:, 0]
        p[:, :k] = c0[:, None] + rng.normal(0, 1, (3, k)) * np.array([[0.3], [0.3], [0.3]])
        p[:, -3:] = np.array([[np.nan, 1e7, 0.0], [0.0, 0.0, np.inf], [0.1, 0.2, 1e9]])
    p[2] = np.where(np.isfinite(p[2]) & (np.abs(p[2]) < 1e6), (p[2] + np.pi) % (2 * np.pi) - np.pi, p[2])
    obs = np.interp(np.linspace(0.0, 1080.0, B), np.arange(1081), full).astype(np.float32)
    obs = np.clip(obs + rng.normal(0, 0.05, B), 0.0, 40.0).astype(np.float32)
    t0 = time.time()
    e = engine.Engine(max_particles=n, seed=int(rng.integers(1, 1 << 30)), max_range_m=max_range)
    try:
        e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
        e.set_beam_angles(ang)
        e.set_particles(p, np.full(n, 1.0 / n))
        e.sensor_update(obs)
        kern = e.ray_kernel_name()
        got = e.log_weights()
        om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y, max_range_m=max_range)
        L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
        fin = np.flatnonzero(np.isfinite(p).all(axis=0) & (np.abs(p[2]) < 1e6))
        pick = rng.choice(fin, min(1500, fin.size), replace=False)
        want, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(p[:, pick]), ang, orc.obs_index(obs, om), L)
        ok1 = np.array_equal(got[pick], want)
        e.update((0.05, 0.0, 0.01), obs)
        q = e.get_particles()
        got2 = e.log_weights()
        fin2 = np.flatnonzero(np.isfinite(q).all(axis=0) & (np.abs(q[2]) < 1e6))
        pick2 = rng.choice(fin2, min(1500, fin2.size), replace=False)
        want2, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(q[:, pick2]), ang, orc.obs_index(obs, om), L)
        ok2 = np.array_equal(got2[pick2], want2)
        c = e.counters()
        variant = e.ray_kernel_variant() if kern == "k_rays_sweep" else {}
    finally:
        e.close()
    bad += (not ok1) + (not ok2)
    if kern == "k_rays_sweep":           # the form of the LAST ray stage: /g global wedge fields, /h hybrid (LDS windows + the global fields beyond)
        kern += "/h" if variant.get("hybrid") else "/g" if variant.get("global_fields") else ""
    print(f"case {case:3d} {mname:9s} range {max_range:4.1f} ({om.max_range_px:3d} px) B {B:4d} span {span:5.2f} a0 {a0:5.2f} {spacing:6s} n {n:6d} {kind:8s} "
          f"{kern:14s} off {c['off_window_particles']:6d} lvl2 {c['level2_rays']:7d} {'OK' if ok1 else 'MISMATCH(sensor)'} {'OK' if ok2 else 'MISMATCH(update)'} "
          f"{time.time() - t0:.1f}s", flush=True)
print("mismatching comparisons:", bad)
sys.exit(1 if bad else 0)
