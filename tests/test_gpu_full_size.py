"""Full-size (BASELINE.json configs[1..3]) checks through size-independent properties, plus a sampled
per-particle comparison with the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_engine

pytestmark = pytest.mark.gpu
ACTION = (0.05, 0.0, 0.01)


@pytest.mark.parametrize("n,mapname", [(262144, "spielberg"), (4194304, "spielberg"), (1048576, "levine")])
def test_full_size_properties(orc, engine_mod, maps_mod, spielberg, n, mapname):
    from monte_carlo_localization_amd import synth
    m = spielberg if mapname == "spielberg" else maps_mod.synthetic_levine()
    pose = (0.0, 0.0, 0.0) if mapname == "spielberg" else (-34.0, -34.9, 0.0)
    ang = synth.beam_angles()
    e = make_engine(engine_mod, m, ang, n, seed=5)
    scan = synth.scan_from_pose(e, m, ang, pose)
    if mapname == "spielberg":
        assert np.array_equal(scan, np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"])
    rng = np.random.default_rng(1)
    p = synth.tracking_cloud(rng, n, pose)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.update(ACTION, scan)
    e.update(ACTION, scan)
    w = e.get_weights()
    idx = e.resample_indices()
    lw = e.log_weights()
    parts = e.get_particles()
    assert np.isfinite(w).all() and (w >= 0).all() and abs(w.sum() - 1.0) < 1e-9
    assert np.isfinite(lw).all() and (lw < 0).all()
    assert idx.min() >= 0 and idx.max() < n
    assert (np.abs(parts[2]) <= np.pi + 1e-12).all()
    # normalised weights == exp(logw - max)/sum, recomputed on the host
    ww = np.exp(lw - lw.max()); ww /= ww.sum()
    np.testing.assert_allclose(w, ww, rtol=1e-9, atol=1e-300)
    # pose == weighted mean of the particles the engine hands back
    pw = e.expected_pose()
    assert abs(pw[0] - (w * parts[0]).sum()) < 1e-9 and abs(pw[1] - (w * parts[1]).sum()) < 1e-9
    assert abs(pw[2] - np.arctan2((w * np.sin(parts[2])).sum(), (w * np.cos(parts[2])).sum())) < 1e-9
    # sampled exact comparison with the oracle: 256 particles' log-weights
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    pick = rng.choice(n, 256, replace=False)
    logw, _, _ = orc.eng_log_weights(om, parts[:, pick], ang, orc.obs_index(scan, om), L)
    assert np.array_equal(lw[pick], logw)
    c = e.counters()
    assert c["level2_rays"] < 0.01 * n * ang.size and c["exact_fallback_rays"] < 1e-5 * n * ang.size


def test_determinism_same_seed_same_bits(orc, engine_mod, spielberg):
    from monte_carlo_localization_amd import synth
    ang = synth.beam_angles(angle_step=4)
    n = 100000
    rng = np.random.default_rng(3)
    p = synth.tracking_cloud(rng, n)
    scan = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::4].copy()
    outs = []
    for rep in range(2):
        e = make_engine(engine_mod, spielberg, ang, n, seed=99)
        e.set_particles(p, np.full(n, 1.0 / n))
        for _ in range(3):
            e.update(ACTION, scan)
        outs.append((e.get_particles(), e.get_weights(), e.resample_indices(), e.expected_pose()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
