#!/usr/bin/env python3
"""The hybrid form of k_rays_sweep on the 479-px stand-in (Spielberg resampled to 0.025 m) against the global-field form (test infrastructure,
run on a GPU box): 1M particles x 1081 beams, N updates (default 150) -- the default choice (global fields for the freshly loaded set, the hybrid
after), MCL_SWEEP_HYBRID=0 and MCL_SWEEP_HYBRID=2 must leave bit-identical particles, weights and resample indices; the last log-weights of the
first run are checked against the oracle.  usage: tools/soak_hybrid.py [updates]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth      # noqa: E402
from oracle import oracle as orc                                  # noqa: E402
sp = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
fine = maps.synthetic_fine025(sp)
ang = synth.beam_angles()
n, steps = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 150
out = {}
for name, env in (("hybrid", {}), ("global-fields", {"MCL_SWEEP_HYBRID": "0"}), ("hybrid-always", {"MCL_SWEEP_HYBRID": "2"})):
    for k, v in env.items(): os.environ[k] = v
    e = engine.Engine(max_particles=n, seed=7)
    e.set_map(fine.data, fine.resolution, fine.origin_x, fine.origin_y); e.set_beam_angles(ang)
    for k in env: os.environ.pop(k, None)
    scan = synth.scan_from_pose(e, fine, ang, (0.0, 0.0, 0.0))
    e.set_particles(synth.tracking_cloud(np.random.default_rng(5), n), np.full(n, 1.0 / n))
    forms = {}
    for k in range(steps):
        e.update((0.05, 0.0, 0.01), scan)
        v = e.ray_kernel_variant(); f = "h" if v["hybrid"] else "g" if v["global_fields"] else "l"
        forms[f] = forms.get(f, 0) + 1
    p, w, idx = e.get_particles(), e.get_weights(), e.resample_indices()
    h = hashlib.sha256(p.tobytes() + w.tobytes() + idx.tobytes()).hexdigest()[:16]
    c = e.counters()
    out[name] = h
    print(f"{name:14s} forms {forms} sha {h} off_window {c['off_window_particles']} level2 {c['level2_rays']}", flush=True)
    if name == "hybrid":
        om = orc.OracleMap(fine.data, fine.resolution, fine.origin_x, fine.origin_y)
        L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
        pick = np.random.default_rng(1).choice(n, 1500, replace=False)
        want, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(p[:, pick]), ang, orc.obs_index(scan, om), L)
        print("   last update's log-weights of 1500 sampled particles vs the oracle:", "EQUAL" if np.array_equal(e.log_weights()[pick], want) else "DIFFERENT", flush=True)
    e.close()
print("identical" if len(set(out.values())) == 1 else "DIFFERENT", flush=True)
sys.exit(0 if len(set(out.values())) == 1 else 1)
