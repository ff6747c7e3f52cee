"""Long runs of the round-3 paths against each other: default (compact list, sweep with atomics) vs MCL_NO_COMPACT + k_rays_cell, bit for bit;
4M x 1081 for 600 updates with an oracle spot check at the end."""
import os, sys, hashlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from monte_carlo_localization_amd import engine, maps, synth
from oracle import oracle as orc
m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
full = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
def run(n, astep, steps, regime, rk, nocompact=False, mode=0):
    ang = synth.beam_angles()[::astep].copy(); scan = full[::astep].copy()
    if nocompact: os.environ["MCL_NO_COMPACT"] = "1"
    e = engine.Engine(max_particles=n, seed=7, ray_kernel=rk, resample_mode=mode)
    os.environ.pop("MCL_NO_COMPACT", None)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y); e.set_beam_angles(ang)
    if regime == "tracking": e.init_particles_pose((0.0, 0.0, 0.0), n)
    else: e.init_global(n)
    rng = np.random.default_rng(5)
    used = 0
    t0 = time.perf_counter()
    for t in range(steps):
        e.update((0.05, 0.0, 0.01), np.clip(scan + rng.normal(0, 0.02, scan.size), 0, 30).astype(np.float32))
        used += e.compact_list()[1]
    dt = time.perf_counter() - t0
    h = hashlib.sha256(e.get_particles().tobytes() + e.get_weights().tobytes() + e.resample_indices().tobytes()).hexdigest()[:16]
    out = (h, [round(float(v), 6) for v in e.expected_pose()], round(dt / steps * 1e3, 3), e.ray_kernel_name(), used, e.counters())
    return out, e
for regime in ("tracking", "global"):
    for mode in (0, 1):
        a, ea = run(1048576, 2, 400, regime, engine.RAYS_AUTO, False, mode); ea.close()
        b, eb = run(1048576, 2, 400, regime, engine.RAYS_CELL, True, mode); eb.close()
        print(regime, "mode", mode, a, "\n     ", b, "IDENTICAL" if a[0] == b[0] else "DIFFERENT", flush=True)
a, e = run(4 << 20, 1, 600, "tracking", engine.RAYS_AUTO)
om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
ang = synth.beam_angles()
parts, lw = e.get_particles(), e.log_weights()
pick = np.random.default_rng(1).choice(4 << 20, 4000, replace=False)
# the last scan the engine saw
rng = np.random.default_rng(5)
for t in range(600): last = np.clip(full + rng.normal(0, 0.02, full.size), 0, 30).astype(np.float32)
L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
want, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts[:, pick]), ang, orc.obs_index(last, om), L)
print("4M x 1081, 600 updates:", a, "oracle spot check:", "EQUAL" if np.array_equal(lw[pick], want) else "MISMATCH", flush=True)
