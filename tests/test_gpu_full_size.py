"""Full-size (BASELINE.json configs[1..3]) checks through size-independent properties, plus a sampled
per-particle comparison with the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_engine, tracking_cloud

pytestmark = pytest.mark.gpu
ACTION = (0.05, 0.0, 0.01)


@pytest.mark.parametrize("n,mapname", [(262144, "spielberg"), (4194304, "spielberg"), (4194304, "levine")])
def test_full_size_properties(orc, engine_mod, maps_mod, spielberg, n, mapname):
    """BASELINE.json configs[1], [2] and [3] at their stated sizes (levine = the synthetic 2049 x 2049 stand-in, the
    reference's maps/levine.pgm is absent).  Two updates from the spread sigma = 0.5 m cloud:
      * the parents of BOTH resampling steps equal the oracle's search of the exact integer CDF (orc_eng_resample_indices on
        the fixed-point weights the spec derives from the log-weights, Philox draws of the spec) for every one of the n children;
      * the log-weights of 16 384 sampled particles after the FIRST update (still the spread cloud) and of 4 096 after the
        second equal the oracle's bit for bit;
      * size-independent properties of the whole set (normalisation, pose = weighted mean, angle range)."""
    from monte_carlo_localization_amd import synth
    m = spielberg if mapname == "spielberg" else maps_mod.synthetic_levine()
    pose = (0.0, 0.0, 0.0) if mapname == "spielberg" else (-34.0, -34.9, 0.0)
    ang = synth.beam_angles()
    seed = 5
    e = make_engine(engine_mod, m, ang, n, seed=seed)
    scan = synth.scan_from_pose(e, m, ang, pose)
    if mapname == "spielberg":
        assert np.array_equal(scan, np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"])
    rng = np.random.default_rng(1)
    p = synth.tracking_cloud(rng, n, pose)
    w0 = np.full(n, 1.0 / n)
    e.set_particles(p, w0)
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(scan, om)

    # ---- first update: uniform weights, spread cloud
    e.update(ACTION, scan)
    assert e.ray_kernel_name() == "k_rays_sweep"
    idx1 = e.resample_indices()
    want1 = orc.eng_resample_indices(orc.eng_quantize_weights(w0), 0, k53=orc.eng_philox_k53(seed, 0, 0, n))
    assert np.array_equal(idx1, want1)
    parts1 = e.get_particles()
    lw1 = e.log_weights()
    pick = rng.choice(n, 16384, replace=False)
    # the children are the oracle's motion model applied to the selected parents with the spec's normals ...
    sub = pick[:2048]
    nrm = orc.eng_philox_normals(seed, 0, 0, n)[sub]
    np.testing.assert_allclose(parts1[:, sub], orc.motion_model(p[:, idx1[sub]], ACTION, nrm), rtol=1e-13, atol=1e-13)
    # ... and their log-weights are the oracle's, on the cloud BEFORE it collapses
    logw, _, _ = orc.eng_log_weights(om, parts1[:, pick], ang, oi, L)
    assert np.array_equal(lw1[pick], logw)
    assert parts1[0].std() > 0.3 and parts1[1].std() > 0.3          # still the sigma = 0.5 m cloud
    del p

    # ---- second update: the peaked weights of the first
    _, q1, _ = orc.eng_weights_from_log(lw1)
    e.update(ACTION, scan)
    idx2 = e.resample_indices()
    want2 = orc.eng_resample_indices(q1, 0, k53=orc.eng_philox_k53(seed, 1, 0, n))
    assert np.array_equal(idx2, want2)
    w = e.get_weights()
    lw = e.log_weights()
    parts = e.get_particles()
    assert np.isfinite(w).all() and (w >= 0).all() and abs(w.sum() - 1.0) < 1e-9
    assert np.isfinite(lw).all() and (lw < 0).all()
    assert (np.abs(parts[2]) <= np.pi + 1e-12).all()
    # normalised weights == exp(logw - max)/sum, recomputed on the host
    ww = np.exp(lw - lw.max()); ww /= ww.sum()
    np.testing.assert_allclose(w, ww, rtol=1e-9, atol=1e-300)
    # pose == weighted mean of the particles the engine hands back
    pw = e.expected_pose()
    assert abs(pw[0] - (w * parts[0]).sum()) < 1e-9 and abs(pw[1] - (w * parts[1]).sum()) < 1e-9
    assert abs(pw[2] - np.arctan2((w * np.sin(parts[2])).sum(), (w * np.cos(parts[2])).sum())) < 1e-9
    pick2 = rng.choice(n, 4096, replace=False)
    logw2, _, _ = orc.eng_log_weights(om, parts[:, pick2], ang, oi, L)
    assert np.array_equal(lw[pick2], logw2)
    c = e.counters()
    assert c["level2_rays"] < 0.01 * n * ang.size and c["exact_fallback_rays"] < 1e-5 * n * ang.size
    e.close()


def test_config_1_4000_particles_1081_beams_through_k_rays_skip(orc, engine_mod, spielberg, spielberg_oracle):
    """BASELINE.json configs[0] on the GPU (the reference's own CPU-runnable case): 4000 x 1081 takes k_rays_skip under AUTO;
    every ray step, log-weight and resample index of two updates equals the oracle."""
    from monte_carlo_localization_amd import synth
    om = spielberg_oracle
    ang = synth.beam_angles()
    n, seed = 4000, 17
    scan = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
    p = synth.tracking_cloud(np.random.default_rng(2), n)
    e = make_engine(engine_mod, spielberg, ang, n, seed=seed, keep_ray_steps=1, graph_mode=1)
    w0 = np.full(n, 1.0 / n)
    e.set_particles(p, w0)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(scan, om)
    q = orc.eng_quantize_weights(w0)
    for upd in range(2):
        e.update(ACTION, scan)
        assert e.ray_kernel_name() == "k_rays_skip"
        assert np.array_equal(e.resample_indices(), orc.eng_resample_indices(q, 0, k53=orc.eng_philox_k53(seed, upd, 0, n)))
        parts = e.get_particles()
        logw, steps, _ = orc.eng_log_weights(om, parts, ang, oi, L, want_steps=True)
        assert np.array_equal(e.ray_steps(), steps)
        assert np.array_equal(e.log_weights(), logw)
        _, q, _ = orc.eng_weights_from_log(logw)
    e.close()


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


def test_nine_million_particles_plan_in_two_passes(orc, engine_mod, spielberg):
    """k_sweep_plan handles 1024 blocks of 8 units per pass: 9 000 000 particles are 8790 units = 1099 blocks, i.e. the
    second pass and its carry.  Every log-weight of k_rays_sweep equals k_rays_cell's (a run the plan lost would leave
    its slots' sums unwritten), and a sample equals the oracle's."""
    n = 9_000_000
    ang = orc.beam_angles(angle_step=18)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].astype(np.float32).copy()
    rng = np.random.default_rng(99)
    p = tracking_cloud(rng, n, sig=(0.4, 0.4, 0.4))
    out = {}
    for name, rk in (("sweep", engine_mod.RAYS_SWEEP), ("cell", engine_mod.RAYS_CELL)):
        e = make_engine(engine_mod, spielberg, ang, n, ray_kernel=rk)
        e.set_particles(p, np.full(n, 1.0 / n))
        e.sensor_update(obs)
        assert e.ray_kernel_name() == "k_rays_" + name
        out[name] = e.log_weights()
        e.close()
    assert np.array_equal(out["sweep"], out["cell"])
    om = orc.OracleMap(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    pick = rng.choice(n, 4096, replace=False)
    T = orc.sensor_table(om.max_range_px)
    logw, _, _ = orc.eng_log_weights(om, p[:, pick], ang, orc.obs_index(obs, om), orc.eng_log_table(T))
    assert np.array_equal(out["sweep"][pick], logw)


def test_fixed_point_cdf_vs_discrete_distribution_at_4m(orc, engine_mod, sibal1):
    """The resampling step's one spec deviation MEASURED at the headline size: 4 194 304 children drawn by the engine
    from the exact integer CDF of weights quantised to 2^-36 of the maximum, against std::discrete_distribution's floating-point
    partial sums (cpp:658-663, orc_ref_resample_indices) under the same injected uniforms, for the weights an update with many beams
    leaves (a few per cent of the particles carry everything).  The two can only disagree where a draw falls within ~N * 2^-36
    (relative) of a CDF step: the count is asserted (and printed for the record), and every disagreement picks a neighbour in CDF
    order (nothing but particles of negligible weight in between)."""
    n = 4194304
    rng = np.random.default_rng(78)
    ang = orc.beam_angles(angle_step=54)
    p = np.vstack([rng.uniform(-2, 2, n), rng.uniform(-1, 1, n), rng.uniform(-np.pi, np.pi, n)])
    w = np.exp(-np.abs(rng.normal(0.0, 40.0, n)))
    w /= w.sum()
    u = rng.random(n)
    e = make_engine(engine_mod, sibal1, ang, n)
    e.set_particles(p, w)
    e.update((0.0, 0.0, 0.0), np.full(ang.size, 3.0, np.float32), uniforms=u)
    got = e.resample_indices()
    e.close()
    want = orc.resample_indices(w, u)
    bad = np.nonzero(got != want)[0]
    k53 = np.minimum((u * 9007199254740992.0).astype(np.uint64), np.uint64(9007199254740991))
    assert np.array_equal(got, orc.eng_resample_indices(orc.eng_quantize_weights(w), 0, k53=k53))     # the engine's own spec: exactly
    print(f"fixed-point CDF vs std::discrete_distribution at {n}: {bad.size} children differ ({bad.size / n:.2e} of the set)")
    # measured on MI355X / this oracle (round 4): 3 of 4 194 304
    assert bad.size <= 64, f"{bad.size} of {n} children differ from std::discrete_distribution"
    q = orc.eng_quantize_weights(w)
    for m in bad:
        # the two picks are neighbours in CDF order up to particles of negligible weight: what lies strictly between them sums to
        # at most 2^-26 of the LARGEST weight (the floating-point partial sums of cpp:658-663 cannot resolve such a step either)
        lo, hi = sorted((int(got[m]), int(want[m])))
        assert int(q[lo + 1:hi].sum()) <= 1024, (m, lo, hi, q[lo:hi + 1])
