"""Stress comparison of MCL_RAYS_QUAD against MCL_RAYS_SKIP on changing inputs (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monte_carlo_localization_amd import engine, maps, synth
m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
step = int(sys.argv[1]) if len(sys.argv) > 1 else 9
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ang = synth.beam_angles(angle_step=step)
scan = np.load(os.path.join(ROOT, "tests/golden/scan_Spielberg_map_origin.npz"))["ranges"][::step].copy()
es = {}
for name, k in (("skip", engine.RAYS_SKIP), ("quad", engine.RAYS_QUAD)):
    e = engine.Engine(max_particles=N, keep_ray_steps=1 if N * ang.size < 2**28 else 0, ray_kernel=k)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y); e.set_beam_angles(ang)
    es[name] = e
rng = np.random.default_rng(3)
bad_total = 0
for rep in range(reps):
    p = synth.tracking_cloud(rng, N, (rng.uniform(-1, 1), rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5))) if rep % 3 else synth.global_cloud(rng, m, N)
    out = {}
    for name, e in es.items():
        e.set_particles(p, np.full(N, 1.0 / N))
        e.sensor_update(scan)
        out[name] = (e.log_weights().copy(), e.counters())
    nb = int((out["quad"][0] != out["skip"][0]).sum())
    bad_total += nb
    print("rep", rep, "logw mismatches", nb, out["quad"][1], flush=True)
print("TOTAL MISMATCHES", bad_total)
# ---- detail: recompute log-weights from the steps the engine reports
sys.path.insert(0, ROOT)
from oracle import oracle as orc
om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
oi = orc.obs_index(scan, om)
p = synth.tracking_cloud(rng, N, (0.3, 0.1, 0.2))
res = {}
for name, e in es.items():
    e.set_particles(p, np.full(N, 1.0 / N)); e.sensor_update(scan)
    st = e.ray_steps().astype(np.int64)
    lw = e.log_weights()
    from_steps = L[oi[None, :], st].astype(np.float64).sum(axis=1)
    res[name] = (st, lw, from_steps)
    print(name, "logw != sum over own steps:", int((lw != from_steps).sum()))
print("steps differ:", int((res["quad"][0] != res["skip"][0]).sum()))
bad = np.nonzero(res["quad"][1] != res["quad"][2])[0]
for i in bad[:10]:
    d = res["quad"][1][i] - res["quad"][2][i]
    row = L[oi, res["quad"][0][i]].astype(np.float64)
    hit = np.nonzero(np.isclose(np.abs(row), abs(d), rtol=0, atol=1e-12))[0]
    print("  particle", i, "diff", d, "matches single table term of beams", hit[:6])
