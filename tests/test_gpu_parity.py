"""GPU parity tests proper: every stage of the update through the C ABI (ctypes -> libmcl_hip_engine.so)
against the CPU oracle and the committed golden fixtures.

Tolerances: integer/index results bit-exact; sensor table bit-exact (fp64);
log-weights bit-exact vs the engine-spec restatement; weights rtol 1e-5 vs the reference product
(<= 121 beams) and vs the log-domain restatement; motion output rtol 1e-13 (device libm vs glibc);
pose within 1 cm / 0.5 deg under identical injected randomness.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_engine, tracking_cloud

pytestmark = pytest.mark.gpu


def KERNELS(engine_mod):
    return {"auto": engine_mod.RAYS_AUTO, "cell": engine_mod.RAYS_CELL, "sweep": engine_mod.RAYS_SWEEP}

ACTION = (0.05, 0.0, 0.01)


def load(name):
    return np.load(os.path.join(GOLDEN, name))


# ------------------------------------------------------------------------------------------- T / H1
@pytest.mark.parametrize("mapname,P", [("spielberg", 207), ("sibal1", 239)])
def test_sensor_table_bit_exact(request, orc, engine_mod, mapname, P):
    m = request.getfixturevalue(mapname)
    e = make_engine(engine_mod, m, orc.beam_angles(angle_step=18), 8)
    assert e.max_range_px == P
    T = e.sensor_table()
    assert np.array_equal(T, orc.sensor_table(P))
    assert np.array_equal(T, load(f"g1_sensor_table_P{P}.npz")["table"])


# ------------------------------------------------------------------------------------------- C (G2)
@pytest.mark.parametrize("name", ["Spielberg_map", "sibal1"])
@pytest.mark.parametrize("kernel", ["march", "skip", "skip_l2", "auto", "cell", "sweep"])
def test_cast_ray_golden(orc, engine_mod, maps_mod, name, kernel):
    """One particle per golden ray, a single beam at angle 0: step index == fixture (incl. rays that
    start outside the map, inside walls, within a cell of the lower/left edge, axis-aligned)."""
    m = maps_mod.load_npz(os.path.join(GOLDEN, f"map_{name}.npz"))
    z = load(f"g2_cast_ray_{name}.npz")
    n = z["x"].size
    rk = {"march": engine_mod.RAYS_MARCH, "auto": engine_mod.RAYS_AUTO, "cell": engine_mod.RAYS_CELL,
          "sweep": engine_mod.RAYS_SWEEP}.get(kernel, engine_mod.RAYS_SKIP)
    e = make_engine(engine_mod, m, np.zeros(1, np.float32), n, keep_ray_steps=1, ray_kernel=rk,
                    debug_force_exact=2 if kernel == "skip_l2" else 0)
    e.set_particles(np.stack([z["x"], z["y"], z["theta"]]), np.full(n, 1.0 / n))
    e.sensor_update(np.array([1.0], np.float32))
    assert np.array_equal(e.ray_steps()[:, 0].astype(np.int16), z["steps"])


@pytest.mark.parametrize("path", ["auto", "cell", "sweep"])
def test_cast_ray_many_beams_scattered_particles(orc, engine_mod, sibal1, sibal1_oracle, path):
    """Particles scattered over the whole (small) map and beyond it: exercises the off-window path."""
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=5)
    rng = np.random.default_rng(11)
    n = 600
    p = np.stack([om.origin_x + rng.uniform(-2, 20, n), om.origin_y + rng.uniform(-2, 11, n), rng.uniform(-np.pi, np.pi, n)])
    e = make_engine(engine_mod, sibal1, ang, n, keep_ray_steps=1, ray_kernel=KERNELS(engine_mod)[path])
    e.set_particles(p, np.full(n, 1.0 / n))
    obs = np.full(ang.size, 3.0, np.float32)
    e.sensor_update(obs)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps(), steps)
    assert np.array_equal(e.log_weights(), logw)


@pytest.mark.parametrize("path", ["auto", "cell", "sweep"])
def test_global_regime_uses_fallback_and_stays_exact(orc, engine_mod, spielberg, spielberg_oracle, path):
    from monte_carlo_localization_amd import synth
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=40)
    rng = np.random.default_rng(12)
    n = 2048
    p = synth.global_cloud(rng, spielberg, n)
    e = make_engine(engine_mod, spielberg, ang, n, keep_ray_steps=1, ray_kernel=KERNELS(engine_mod)[path])
    e.set_particles(p, np.full(n, 1.0 / n))
    obs = load("scan_Spielberg_map_origin.npz")["ranges"][::40].copy()
    e.sensor_update(obs)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps(), steps)
    assert np.array_equal(e.log_weights(), logw)
    if path == "auto":       # the cell-sorted slices are compact: hardly any pair misses its window there
        assert e.counters()["off_window_particles"] > 0


@pytest.mark.parametrize("path", ["auto", "cell", "sweep"])
def test_nonfinite_particles_do_not_hang_or_crash(orc, engine_mod, sibal1, path):
    ang = orc.beam_angles(angle_step=60)
    n = 64
    p = np.zeros((3, n))
    p[0, 0] = np.nan; p[1, 1] = np.inf; p[0, 2] = -np.inf; p[2, 3] = np.nan; p[0, 4] = 1e300; p[2, 5] = 1e300
    e = make_engine(engine_mod, sibal1, ang, n, keep_ray_steps=1, ray_kernel=KERNELS(engine_mod)[path])
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(np.full(ang.size, 2.0, np.float32))
    assert e.ray_steps().shape == (n, ang.size)
    # the finite particles are still exact, and a huge-but-finite heading is marched literally like the oracle does
    om = orc.OracleMap(sibal1.data, sibal1.resolution, sibal1.origin_x, sibal1.origin_y)
    fin = np.array([5] + list(range(6, n)))
    obs = np.full(ang.size, 2.0, np.float32)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    _, steps, _ = orc.eng_log_weights(om, p[:, fin], ang, orc.obs_index(obs, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps()[fin], steps)
    e.update(ACTION, np.full(ang.size, 2.0, np.float32))     # motion + normalize_angle on garbage must terminate


# ------------------------------------------------------------------------------------------- E / D4
@pytest.mark.parametrize("B", [61, 121])
def test_weights_match_reference_product(orc, engine_mod, spielberg, spielberg_oracle, B):
    z = load(f"g3_mcl_step_B{B}.npz")
    n = z["particles_out"].shape[1]
    for mode, rtol in ((engine_mod.WEIGHT_LOG, 1e-5), (engine_mod.WEIGHT_PRODUCT, 1e-12)):
        e = make_engine(engine_mod, spielberg, z["angles"], n, keep_ray_steps=1, weight_mode=mode)
        e.set_particles(z["particles_out"], np.full(n, 1.0 / n))
        e.sensor_update(z["obs"])
        assert np.array_equal(e.ray_steps(), z["steps"])
        np.testing.assert_allclose(e.get_weights(), z["weights_out"], rtol=rtol, atol=0)
        pose = e.expected_pose()
        np.testing.assert_allclose(pose, z["pose"], rtol=0, atol=1e-6)


def test_underflow_witness_1081_beams(orc, engine_mod, spielberg):
    """G4 / SURVEY D4: product mode reproduces the reference's all-zero weights bit for bit; log mode
    stays finite and equals the log-domain restatement."""
    z = load("g4_underflow_B1081.npz")
    ang = orc.beam_angles()
    obs = load("scan_Spielberg_map_origin.npz")["ranges"]
    n = z["particles_out"].shape[1]
    e = make_engine(engine_mod, spielberg, ang, n, keep_ray_steps=1, weight_mode=engine_mod.WEIGHT_PRODUCT)
    e.set_particles(z["particles_out"], np.full(n, 1.0 / n))
    e.sensor_update(obs)
    assert np.array_equal(e.ray_steps(), z["steps"])
    assert np.array_equal(e.get_weights(), z["ref_weights"]) and e.get_weights().sum() == 0.0
    # and the next resample degenerates exactly like the reference: every index 0
    e.update(ACTION, obs, normals=np.zeros((n, 3)), uniforms=np.linspace(0.01, 0.99, n))
    assert (e.resample_indices() == 0).all()
    e2 = make_engine(engine_mod, spielberg, ang, n, keep_ray_steps=1)
    e2.set_particles(z["particles_out"], np.full(n, 1.0 / n))
    e2.sensor_update(obs)
    assert np.array_equal(e2.log_weights(), z["eng_logw"])
    w, q, mx = orc.eng_weights_from_log(z["eng_logw"])
    np.testing.assert_allclose(e2.get_weights(), w / w.sum(), rtol=1e-13)
    assert np.isfinite(e2.get_weights()).all() and abs(e2.get_weights().sum() - 1.0) < 1e-12


def test_observation_edge_cases(orc, engine_mod, sibal1, sibal1_oracle):
    """G6: obs = +inf, > max range, NaN, negative."""
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=120)
    obs = np.array([np.inf, 100.0, np.nan, -3.0, 0.0, 5.0, 11.99, 12.0, 0.024, 0.026], np.float32)[: ang.size]
    assert list(orc.obs_index(obs, om)[:5]) == [om.max_range_px, om.max_range_px, 0, 0, 0]
    rng = np.random.default_rng(4)
    n = 200
    p = tracking_cloud(rng, n, (2.0, 2.0, 0.3), (0.3, 0.3, 0.3))
    e = make_engine(engine_mod, sibal1, ang, n)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(obs)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, _, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L)
    assert np.array_equal(e.log_weights(), logw)


# ------------------------------------------------------------------------------------------- M
@pytest.mark.parametrize("action", [(0.05, 0.0, 0.01), (0.3, 0.0, -0.2), (0.0005, 0.0, 0.0005), (0.0, 0.0, 0.05),
                                    (-0.08, 0.0, 0.0)])
def test_motion_model_injected_normals(orc, engine_mod, sibal1, action):
    rng = np.random.default_rng(7)
    n = 1000
    p = tracking_cloud(rng, n, (2.0, 2.0, 3.0), (0.3, 0.3, 0.5))
    nrm = rng.normal(size=(n, 3))
    nrm[:10, 2] = 20.0          # forces several +-2pi wraps
    ang = orc.beam_angles(angle_step=120)
    e = make_engine(engine_mod, sibal1, ang, n)
    w = np.zeros(n); w[:] = 1.0 / n
    e.set_particles(p, w)
    u = (np.arange(n) + 0.5) / n            # uniform weights + these uniforms -> identity resample
    e.update(action, np.full(ang.size, 2.0, np.float32), normals=nrm, uniforms=u)
    assert np.array_equal(e.resample_indices(), np.arange(n))
    want = orc.motion_model(p, action, nrm)
    got = e.get_particles()
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)
    assert (np.abs(got[2]) <= np.pi + 1e-12).all()


# ------------------------------------------------------------------------------------------- R
@pytest.mark.parametrize("B", [61, 121])
def test_full_step_golden_g3(orc, engine_mod, spielberg, B):
    z = load(f"g3_mcl_step_B{B}.npz")
    n = z["particles_in"].shape[1]
    e = make_engine(engine_mod, spielberg, z["angles"], n, keep_ray_steps=1)
    e.set_particles(z["particles_in"], z["weights_in"])
    np.testing.assert_allclose(e.get_weights(), z["weights_in"], rtol=1e-14)
    e.update(z["action"], z["obs"], normals=z["normals"], uniforms=z["uniforms"])
    assert np.array_equal(e.resample_indices(), z["idx"])                       # bit-exact parents
    np.testing.assert_allclose(e.get_particles(), z["particles_out"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(e.ray_steps(), z["steps"])                            # bit-exact ray steps
    np.testing.assert_allclose(e.get_weights(), z["weights_out"], rtol=1e-5)
    pose = e.expected_pose()
    assert abs(pose[0] - z["pose"][0]) < 1e-6 and abs(pose[1] - z["pose"][1]) < 1e-6 and abs(pose[2] - z["pose"][2]) < 1e-6
    t = e.stage_timings()
    assert t.shape == (6,) and t[5] > 0 and t[3] > 0


def test_resample_indices_bit_exact_native_philox(orc, engine_mod, sibal1):
    """No injection: Philox uniforms + exact integer CDF, multinomial and systematic, vs the scalar
    restatement (orc_eng_*), from arbitrary weights."""
    rng = np.random.default_rng(9)
    n = 5000
    ang = orc.beam_angles(angle_step=120)
    p = tracking_cloud(rng, n, (2.0, 2.0, 0.0), (0.2, 0.2, 0.2))
    w = rng.random(n) ** 6
    w[rng.integers(0, n, 300)] = 0.0
    obs = np.full(ang.size, 2.0, np.float32)
    for mode in (engine_mod.RESAMPLE_MULTINOMIAL, engine_mod.RESAMPLE_SYSTEMATIC):
        e = make_engine(engine_mod, sibal1, ang, n, seed=0xDEADBEEF1234, resample_mode=mode)
        e.set_particles(p, w)
        q = orc.eng_quantize_weights(w)
        for upd in range(2):
            if upd == 1:
                _, q, _ = orc.eng_weights_from_log(e.log_weights())
            e.update(ACTION, obs)
            if mode == engine_mod.RESAMPLE_MULTINOMIAL:
                want = orc.eng_resample_indices(q, 0, k53=orc.eng_philox_k53(0xDEADBEEF1234, upd, 0, n))
            else:
                want = orc.eng_resample_indices(q, 1, k0=orc.eng_philox_k0(0xDEADBEEF1234, upd))
            assert np.array_equal(e.resample_indices(), want), (mode, upd)


def test_native_philox_motion_noise(orc, engine_mod, sibal1):
    rng = np.random.default_rng(10)
    n = 4096
    ang = orc.beam_angles(angle_step=120)
    p = tracking_cloud(rng, n, (2.0, 2.0, 0.0), (0.2, 0.2, 0.2))
    e = make_engine(engine_mod, sibal1, ang, n, seed=77)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.update(ACTION, np.full(ang.size, 2.0, np.float32))
    idx = e.resample_indices()
    nrm = orc.eng_philox_normals(77, 0, 0, n)
    want = orc.motion_model(p[:, idx], ACTION, nrm)
    np.testing.assert_allclose(e.get_particles(), want, rtol=1e-12, atol=1e-12)
    assert abs(nrm.mean()) < 0.05 and abs(nrm.std() - 1.0) < 0.05


@pytest.mark.parametrize("shape", ["flat", "peaked"])
def test_fixed_point_cdf_vs_discrete_distribution_at_262144(orc, engine_mod, sibal1, shape):
    """The one spec deviation of the resampling step, measured: the engine draws from the EXACT integer CDF
    of weights quantised to 2^-36 of the maximum, the reference from std::discrete_distribution's floating-point partial
    sums (cpp:658-663, restated by orc_ref_resample_indices).  Under the same injected uniforms the two disagree only
    where a draw falls within ~N * 2^-36 (relative) of a CDF step -- the reference's own sequential partial sums carry a
    rounding error of that order too: expected O(N^2 * 2^-36 / sum(w / w_max)) children per update, i.e. a handful at
    262 144 particles.  A disagreement picks a neighbour in CDF order (the next particle with weight), never a far one."""
    n = 262144
    rng = np.random.default_rng(77)
    ang = orc.beam_angles(angle_step=54)
    p = np.vstack([rng.uniform(-2, 2, n), rng.uniform(-1, 1, n), rng.uniform(-np.pi, np.pi, n)])
    if shape == "flat":
        w = rng.random(n) + 0.05
    else:                                   # what an update leaves: a few per cent of the particles carry all the weight
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
    # the engine's own spec is met exactly
    k53 = np.minimum((u * 9007199254740992.0).astype(np.uint64), np.uint64(9007199254740991))
    assert np.array_equal(got, orc.eng_resample_indices(orc.eng_quantize_weights(w), 0, k53=k53))
    assert bad.size <= 16, f"{bad.size} of {n} children differ from std::discrete_distribution"
    q = orc.eng_quantize_weights(w)
    for m in bad:                           # a neighbour in CDF order: no particle with fixed-point weight in between
        lo, hi = sorted((int(got[m]), int(want[m])))
        assert not q[lo + 1:hi].any()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", [20000, 300000])
def test_compact_parent_list_equals_full_cdf(orc, engine_mod, spielberg, spielberg_oracle, monkeypatch, n, mode):
    """From the second update on the resampling draws from the compact list of the particles that carry weight (a few per
    cent of the set).  Indices, children and weights equal an engine with the list disabled (MCL_NO_COMPACT=1: full CDF,
    packed records) bit for bit, and the indices equal the oracle's exact-CDF draw from the oracle's own weights."""
    ang = orc.beam_angles(angle_step=2)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::2].astype(np.float32).copy()
    p = tracking_cloud(np.random.default_rng(12), n)
    w0 = np.full(n, 1.0 / n)
    eng = {}
    for tag in ("list", "full"):
        if tag == "full":
            monkeypatch.setenv("MCL_NO_COMPACT", "1")
        eng[tag] = make_engine(engine_mod, spielberg, ang, n, seed=21, resample_mode=mode)
        eng[tag].set_particles(p, w0)
    monkeypatch.delenv("MCL_NO_COMPACT")
    L = orc.eng_log_table(orc.sensor_table(spielberg_oracle.max_range_px))
    oi = orc.obs_index(obs, spielberg_oracle)
    q = orc.eng_quantize_weights(w0)
    prev, times_used = -1, 0
    for upd in range(4):
        for e in eng.values():
            e.update(ACTION, obs)
        n_list, used = eng["list"].compact_list()
        assert used == (prev > 0)
        times_used += used
        prev = n_list
        assert eng["full"].compact_list() == (-1, False)
        idx = eng["list"].resample_indices()
        assert np.array_equal(idx, eng["full"].resample_indices())
        assert np.array_equal(idx, orc.eng_resample_indices(q, mode, k53=orc.eng_philox_k53(21, upd, 0, n), k0=orc.eng_philox_k0(21, upd)))
        parts = eng["list"].get_particles()
        assert np.array_equal(parts, eng["full"].get_particles())
        assert np.array_equal(eng["list"].get_weights(), eng["full"].get_weights())
        logw, _, _ = orc.eng_log_weights(spielberg_oracle, parts, ang, oi, L)
        assert np.array_equal(eng["list"].log_weights(), logw)
        _, q, _ = orc.eng_weights_from_log(logw)
        alive = int(np.count_nonzero(q))
        assert n_list == (alive if alive <= max(4096, (n // 4 + 63) // 64 * 64) else -1)
    assert times_used >= 2
    # a flat weight set (every particle carries weight) has no list: the next update draws from the full CDF again
    e = eng["list"]
    e.set_particles(p, w0)
    assert e.compact_list()[0] == -1
    e.update(ACTION, obs)
    assert e.compact_list()[1] is False
    assert np.array_equal(e.resample_indices(), orc.eng_resample_indices(orc.eng_quantize_weights(w0), mode, k53=orc.eng_philox_k53(21, 4, 0, n),
                                                                       k0=orc.eng_philox_k0(21, 4)))
    for e in eng.values():
        e.close()


def test_all_zero_weights_resample_like_reference(orc, engine_mod, sibal1):
    n = 256
    ang = orc.beam_angles(angle_step=120)
    rng = np.random.default_rng(2)
    p = tracking_cloud(rng, n, (2.0, 2.0, 0.0), (0.2, 0.2, 0.2))
    e = make_engine(engine_mod, sibal1, ang, n)
    e.set_particles(p, np.zeros(n))
    u = rng.random(n)
    e.update(ACTION, np.full(ang.size, 2.0, np.float32), normals=np.zeros((n, 3)), uniforms=u)
    with np.errstate(all="ignore"):
        assert np.array_equal(e.resample_indices(), orc.resample_indices(np.zeros(n), u))   # all 0


# ------------------------------------------------------------------------------------------- D (G5)
def test_closed_loop_trajectory_g5(orc, engine_mod, spielberg, spielberg_oracle):
    """30 updates, N=2000, B=61, identical injected randomness: pose within 1 cm / 0.5 deg of the
    reference chain at every step (BASELINE.json north_star)."""
    z = load("g5_trajectory_N2000_B61.npz")
    om = spielberg_oracle
    a = orc.beam_angles(angle_step=18)
    s = orc.RefStream(42)
    N = 2000
    p, w = orc.init_particles_pose(s, (0.0, 0.0, 0.0), N)
    e = make_engine(engine_mod, spielberg, a, N)
    e.set_particles(p, w)
    worst = np.zeros(3)
    for k in range(30):
        pose_true = z["truth"][k]
        obs, _ = orc.cast_many(om, np.full(a.size, pose_true[0]), np.full(a.size, pose_true[1]), pose_true[2] + a.astype(np.float64))
        u, nrm = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
        e.update(ACTION, obs, normals=nrm, uniforms=u)
        worst = np.maximum(worst, np.abs(e.expected_pose() - z["poses"][k]))
    assert worst[0] < 0.01 and worst[1] < 0.01 and worst[2] < np.deg2rad(0.5), worst
    np.testing.assert_allclose(e.get_particles(), z["final_particles"], rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------------------------------- misc ABI
def test_sample_particles_and_mean(orc, engine_mod, sibal1):
    rng = np.random.default_rng(3)
    n = 3000
    ang = orc.beam_angles(angle_step=120)
    p = tracking_cloud(rng, n, (2.0, 2.0, 0.0), (0.2, 0.2, 0.2))
    w = rng.random(n)
    e = make_engine(engine_mod, sibal1, ang, n)
    e.set_particles(p, w)
    u = rng.random(60)
    got = e.sample_particles(60, u)                                   # visualize(): cpp:949-956
    idx = orc.resample_indices(w, u) if False else orc.eng_resample_indices(
        orc.eng_quantize_weights(w), 0, n_children=60, k53=np.floor(u * 2.0 ** 53).astype(np.uint64))
    assert np.array_equal(got, p[:, idx])
    np.testing.assert_allclose(e.particle_mean(), p.mean(axis=1), rtol=1e-12)
    np.testing.assert_allclose(e.get_weights(), w / w.sum(), rtol=1e-13)
    np.testing.assert_allclose(e.expected_pose(), orc.expected_pose(p, w / w.sum()), atol=1e-12)
    assert e.sample_particles(60).shape == (3, 60)


def test_error_codes(orc, engine_mod, sibal1):
    e = engine_mod.Engine(max_particles=16)
    with pytest.raises(engine_mod.EngineError):       # nothing set yet
        e.sensor_update(np.zeros(4, np.float32))
    e.set_map(sibal1.data, sibal1.resolution, sibal1.origin_x, sibal1.origin_y)
    e.set_beam_angles(np.zeros(4, np.float32))
    with pytest.raises(engine_mod.EngineError):       # too many particles
        e.set_particles(np.zeros((3, 17)), np.ones(17))
    e.set_particles(np.zeros((3, 16)), np.ones(16))
    with pytest.raises(engine_mod.EngineError):       # beam count mismatch
        e.sensor_update(np.zeros(5, np.float32))
    with pytest.raises(engine_mod.EngineError):       # steps not kept
        e.ray_steps()
    with pytest.raises(engine_mod.EngineError):       # invalid resolution (cpp:236-240)
        e.set_map(sibal1.data, 0.0, 0.0, 0.0)


# ------------------------------------------------------------------------------------------- §8(f) next rows
def test_device_init_global_bit_exact(orc, engine_mod, spielberg, spielberg_oracle, sibal1, sibal1_oracle):
    """initialize_global on the device (cpp:401-446 with Philox draws): cell choice, position arithmetic
    and angle are integer/one-rounding operations -> bit-exact vs the scalar restatement, incl. a sharded
    call (first_global_index)."""
    for m, om in ((spielberg, spielberg_oracle), (sibal1, sibal1_oracle)):
        n = 10000
        e = make_engine(engine_mod, m, orc.beam_angles(angle_step=120), n, seed=31337)
        e.init_global(n)
        want = orc.eng_init_global(31337, 0, om, 0, n)
        got = e.get_particles()
        assert np.array_equal(got, want)
        assert np.array_equal(e.get_weights(), np.full(n, 1.0 / n))
        # every particle sits on the lower-left corner of a free cell (cpp:438-439)
        col = np.rint((got[0] - om.origin_x) / om.resolution).astype(int)
        row = np.rint((got[1] - om.origin_y) / om.resolution).astype(int)
        assert (m.data[row, col] == 0).all() and (got[2] >= 0).all() and (got[2] < 2 * np.pi).all()
        e.init_global(n // 2, first_global_index=n // 2, n_total=n)        # second call: init counter 1, upper shard
        want2 = orc.eng_init_global(31337, 1, om, n // 2, n // 2)
        assert np.array_equal(e.get_particles(), want2)
        assert np.array_equal(e.get_weights(), np.full(n // 2, 2.0 / n))   # shard-normalised view of 1/n_total


def test_device_init_pose(orc, engine_mod, sibal1):
    n = 20000
    e = make_engine(engine_mod, sibal1, orc.beam_angles(angle_step=120), n, seed=99)
    pose = (2.0, 1.5, 3.0)
    e.init_particles_pose(pose, n)
    want = orc.eng_init_pose(99, 0, pose, 0, n)
    got = e.get_particles()
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)        # Box-Muller: device libm vs glibc
    assert abs(got[0].std() - 0.5) < 0.02 and abs(got[1].std() - 0.5) < 0.02 and (np.abs(got[2]) <= np.pi).all()
    np.testing.assert_allclose(e.expected_pose()[:2], got[:2].mean(axis=1), atol=1e-12)
    e.update((0.05, 0.0, 0.01), np.full(e.n_beams, 2.0, np.float32))   # usable state


def test_update_scan_downsamples_like_lidarcb(orc, engine_mod, spielberg):
    """mcl_update_scan(raw 1081 ranges, angle_step) == mcl_update(ranges[::angle_step]) (cpp:316-320)."""
    raw = load("scan_Spielberg_map_origin.npz")["ranges"].copy()
    raw[5] = np.inf; raw[23] = np.nan; raw[90] = 40.0
    rng = np.random.default_rng(8)
    n = 3000
    p = tracking_cloud(rng, n)
    outs = []
    for use_raw in (False, True):
        e = make_engine(engine_mod, spielberg, orc.beam_angles(angle_step=18), n, seed=5)
        e.set_particles(p, np.full(n, 1.0 / n))
        if use_raw:
            e.update_scan(ACTION, raw, 18)
        else:
            e.update(ACTION, raw[::18].copy())
        outs.append((e.get_particles(), e.get_weights(), e.log_weights()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # ... and against the ORACLE: lidarCB keeps ranges[i * ANGLE_STEP] (cpp:316-320); the log-weights of the particles the raw-scan
    # update produced equal orc_eng_log_weights on them with that downsampled scan (NaN / +inf / beyond-range readings included)
    om = orc.OracleMap(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    ang18 = orc.beam_angles(angle_step=18)
    down = np.array([raw[i * 18] for i in range(ang18.size)], np.float32)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, _, _ = orc.eng_log_weights(om, outs[1][0], ang18, orc.obs_index(down, om), L)
    assert np.array_equal(outs[1][2], logw)
    e = make_engine(engine_mod, spielberg, orc.beam_angles(angle_step=18), n)
    e.set_particles(p, np.full(n, 1.0 / n))
    with pytest.raises(engine_mod.EngineError):
        e.update_scan(ACTION, raw, 17)                 # beam count mismatch


@pytest.mark.parametrize("mapname", ["Spielberg_map", "sibal1", "icra_2_clean", "first_map"])
@pytest.mark.parametrize("max_range,n_beams_step", [(12.0, 7), (5.0, 13), (3.3, 31)])
@pytest.mark.parametrize("path", ["auto", "cell", "sweep"])
def test_ray_steps_all_maps_and_ranges(orc, engine_mod, maps_mod, mapname, max_range, n_beams_step, path):
    """Every fixture map, three MAX_RANGE_PX values, global clouds (free cells) plus particles inside walls
    and outside the map: steps and log-weights bit-exact vs the oracle through the default kernel."""
    m = maps_mod.load_npz(os.path.join(GOLDEN, f"map_{mapname}.npz"))
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y, max_range_m=max_range)
    from monte_carlo_localization_amd import synth
    ang = orc.beam_angles(angle_step=n_beams_step)
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{mapname}{n_beams_step}".encode()))
    n = 1500
    p = synth.global_cloud(rng, m, n)
    p[:2, :100] += rng.normal(0, 3.0, (2, 100))           # some land in walls / outside the map
    e = make_engine(engine_mod, m, ang, n, keep_ray_steps=1, max_range_m=max_range, squash_factor=3.1,
                    ray_kernel=KERNELS(engine_mod)[path])
    assert e.max_range_px == om.max_range_px
    e.set_particles(p, np.full(n, 1.0 / n))
    obs = rng.uniform(0.0, max_range * 1.2, ang.size).astype(np.float32)
    e.sensor_update(obs)
    T = orc.sensor_table(om.max_range_px)
    assert np.array_equal(e.sensor_table(), T)
    L = orc.eng_log_table(T, 1.0 / 3.1)
    logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps(), steps)
    assert np.array_equal(e.log_weights(), logw)


@pytest.mark.parametrize("path", ["auto", "cell", "sweep"])
def test_tight_cluster_near_map_corner(orc, engine_mod, sibal1, sibal1_oracle, path):
    """Window partly outside the map (lower-left corner): out-of-map cells must read as stops, and the
    truncation-toward-zero column/row (pixel coordinates in (-1,0)) must read cell 0."""
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=3)
    rng = np.random.default_rng(21)
    n = 2000
    p = np.stack([om.origin_x + rng.uniform(-0.2, 1.0, n), om.origin_y + rng.uniform(-0.2, 1.0, n), rng.uniform(-np.pi, np.pi, n)])
    e = make_engine(engine_mod, sibal1, ang, n, keep_ray_steps=1, ray_kernel=KERNELS(engine_mod)[path])
    e.set_particles(p, np.full(n, 1.0 / n))
    obs = np.full(ang.size, 1.0, np.float32)
    e.sensor_update(obs)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps(), steps)
    # k_rays_sweep's windows are 256 cells wide: at this map's 240-px range they leave 11 cells of play, less than the
    # cluster's 24 cells, and the rest of the cluster goes through k_rays_far (same steps, checked above)
    if path != "sweep":
        assert e.counters()["off_window_particles"] == 0


# ------------------------------------------------------------------------------------------- cell-sorted path
def test_cell_kernel_full_turn_scan_and_identical_particles(orc, engine_mod, sibal1, sibal1_oracle):
    """MCL_RAYS_CELL corner cases: a scan of almost a full turn (a wedge is visited twice: two beam ranges per
    lane), many particles in one bucket of the sort (identical poses) and a large spread of the rest."""
    om = sibal1_oracle
    ang = np.linspace(-np.pi, np.pi - 0.004, 720).astype(np.float32)
    rng = np.random.default_rng(31)
    n = 3000
    p = np.stack([rng.uniform(-1.0, 1.0, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-np.pi, np.pi, n)])
    p[:, :1000] = np.array([[0.3], [0.1], [0.7]])                       # one hot bucket
    p[2, 1000:1100] = np.pi * rng.integers(-4, 5, 100) / 8.0             # headings exactly on wedge edges
    e = make_engine(engine_mod, sibal1, ang, n, keep_ray_steps=1, ray_kernel=engine_mod.RAYS_CELL)
    e.set_particles(p, np.full(n, 1.0 / n))
    obs = rng.uniform(0.5, 8.0, ang.size).astype(np.float32)
    e.sensor_update(obs)
    assert e.ray_kernel_name() == "k_rays_cell"
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps(), steps)
    assert np.array_equal(e.log_weights(), logw)


def test_cell_and_quad_agree_over_updates(orc, engine_mod, spielberg):
    """Same seed, 200k particles, thirty updates: k_rays_sweep (default at this size), k_rays_cell and k_rays_quad give
    the same particles, weights and pose bit for bit (the sort only changes which rays share a wave; a lost or
    duplicated slot of the per-XCD counting sort, or a partial sum read back stale, would show up here)."""
    from monte_carlo_localization_amd import synth
    ang = orc.beam_angles(angle_step=9)
    obs = load("scan_Spielberg_map_origin.npz")["ranges"][::9].copy()
    n = 200000
    out = {}
    for name, rk in (("auto", engine_mod.RAYS_AUTO), ("cell", engine_mod.RAYS_CELL), ("quad", engine_mod.RAYS_QUAD)):
        e = make_engine(engine_mod, spielberg, ang, n, ray_kernel=rk, seed=5)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        for _ in range(30):
            e.update(ACTION, obs)
        out[name] = (e.ray_kernel_name(), e.get_particles(), e.get_weights(), e.expected_pose())
    assert out["auto"][0] == "k_rays_sweep" and out["cell"][0] == "k_rays_cell" and out["quad"][0] == "k_rays_quad"
    for other in ("cell", "quad"):
        assert np.array_equal(out["auto"][1], out[other][1])
        assert np.array_equal(out["auto"][2], out[other][2])
        assert np.array_equal(out["auto"][3], out[other][3])


# ------------------------------------------------------------------------------------------- adaptive resampling
def test_adaptive_resampling_neff_option(orc, engine_mod, spielberg, spielberg_oracle):
    """resample_neff_permille = 20 (the reference resamples unconditionally, cpp:656-665): an update
    keeps its particles while the previous weights have N_eff >= 0.02 N and their log-weights add; once N_eff drops
    below, it resamples (at 121 beams one update leaves N_eff at about 0.025 N, so the branches alternate).  Particles and total log-weights bit-exact vs the scalar restatement, both branches taken."""
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=9)
    obs = load("scan_Spielberg_map_origin.npz")["ranges"][::9].copy()
    n, seed = 3000, 11
    rng = np.random.default_rng(4)
    p = tracking_cloud(rng, n, sig=(0.03, 0.03, 0.01))
    e = make_engine(engine_mod, spielberg, ang, n, seed=seed, resample_neff_permille=20)
    e.set_particles(p, np.full(n, 1.0 / n))
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(obs, om)
    w = np.full(n, 1.0 / n)
    q = orc.eng_quantize_weights(w)
    carry = None
    kept = []
    for upd in range(8):
        neff = w.sum() ** 2 / (w * w).sum()
        keep = carry is not None and neff >= 0.02 * n
        assert carry is None or abs(neff / n - 0.02) > 1e-6            # the test input stays away from the threshold
        if keep:
            parents = p
        else:
            idx = orc.eng_resample_indices(q, 0, n_children=n, k53=orc.eng_philox_k53(seed, upd, 0, n))
            parents = p[:, idx]
        p_want = orc.motion_model(parents, ACTION, orc.eng_philox_normals(seed, upd, 0, n))
        e.update(ACTION, obs)
        got_neff, resampled = e.effective_sample_size()
        assert resampled == (not keep), upd
        p = e.get_particles()                    # device sin/cos differ from libm in the last ulp: continue from the engine's set
        np.testing.assert_allclose(p, p_want, rtol=1e-12, atol=1e-12)
        if not keep:
            assert np.array_equal(e.resample_indices(), idx), upd
        logw, _, _ = orc.eng_log_weights(om, p, ang, oi, L)
        if keep:
            logw = logw + carry
        carry = logw - logw.max()
        w = orc.eng_det_exp(carry)
        q = np.floor(w * 2.0 ** 36).astype(np.uint64)
        assert np.array_equal(e.log_weights(), logw), upd
        np.testing.assert_allclose(e.get_weights(), w / w.sum(), rtol=1e-12, atol=0)
        np.testing.assert_allclose(got_neff, w.sum() ** 2 / (w * w).sum(), rtol=1e-12)
        kept.append(keep)
    assert any(kept) and not all(kept[1:]), kept


def test_cell_kernel_irregular_beam_angles(orc, engine_mod, sibal1, sibal1_oracle):
    """Beam angles that increase but are not evenly spaced (gaps, clusters, a few nearly equal): the closed-form guess
    for a wedge's first beam is useless here and the bisection fallback has to find the ranges."""
    om = sibal1_oracle
    rng = np.random.default_rng(41)
    ang = np.sort(np.concatenate([rng.uniform(-2.8, -1.0, 150), rng.uniform(-0.2, -0.19, 40), rng.uniform(0.5, 2.9, 210),
                                  np.array([-np.pi / 2, 0.0, np.pi / 8, np.pi / 2])])).astype(np.float32)
    ang = np.unique(ang)
    n = 2500
    p = np.stack([rng.uniform(-1.0, 1.0, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-np.pi, np.pi, n)])
    for rk in (engine_mod.RAYS_CELL, engine_mod.RAYS_QUAD):
        e = make_engine(engine_mod, sibal1, ang, n, keep_ray_steps=1, ray_kernel=rk)
        e.set_particles(p, np.full(n, 1.0 / n))
        obs = rng.uniform(0.5, 8.0, ang.size).astype(np.float32)
        e.sensor_update(obs)
        L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
        logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
        assert np.array_equal(e.ray_steps(), steps)
        assert np.array_equal(e.log_weights(), logw)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_maps_every_kernel(orc, engine_mod, maps_mod, seed):
    """Randomised geometry: thin and diagonal walls, isolated occupied cells, unknown patches, obstacles on the map
    border; random resolution, MAX_RANGE, beam set and particle spread (inside walls and outside the map included).
    Every ray kernel returns the oracle's steps and log-weights."""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(180, 330)), int(rng.integers(180, 330))
    g = np.zeros((H, W), np.int8)
    for _ in range(int(rng.integers(4, 14))):                      # axis-aligned and diagonal wall segments
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        L = int(rng.integers(10, 120)); dx, dy = [(1, 0), (0, 1), (1, 1), (1, -1), (2, 1), (1, 3)][int(rng.integers(0, 6))]
        for k in range(L):
            x, y = x0 + (k * dx) // max(abs(dx), abs(dy)), y0 + (k * dy) // max(abs(dx), abs(dy))
            if 0 <= x < W and 0 <= y < H:
                g[y, x] = 100
    g[rng.random(g.shape) < 0.004] = 100                           # isolated cells
    g[rng.random(g.shape) < 0.01] = -1                             # unknown: transparent (cpp:642 tests > 50 only)
    if seed % 2:
        g[0, :] = 100; g[:, -1] = 100                              # walls on two borders
    res = np.float32([0.05, 0.05796, 0.1, 0.043][seed % 4])
    m = maps_mod.OccupancyMap(g, res, float(rng.uniform(-20, 5)), float(rng.uniform(-20, 5)), f"rand{seed}")
    max_range = float(rng.uniform(4.0, min(12.0, 250 * float(res))))
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y, max_range_m=max_range)
    B = int(rng.integers(40, 140))
    ang = np.unique(np.sort(rng.uniform(-2.6, 2.6, B)).astype(np.float32))
    n = 1500
    cx, cy = m.origin_x + W * float(res) * rng.uniform(0.2, 0.8), m.origin_y + H * float(res) * rng.uniform(0.2, 0.8)
    spread = float(rng.choice([0.3, 2.0, 8.0]))
    p = np.stack([cx + rng.normal(0, spread, n), cy + rng.normal(0, spread, n), rng.uniform(-np.pi, np.pi, n)])
    obs = rng.uniform(0.0, max_range * 1.1, ang.size).astype(np.float32)
    T = orc.sensor_table(om.max_range_px)
    L = orc.eng_log_table(T)
    logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L, want_steps=True)
    for rk in (engine_mod.RAYS_CELL, engine_mod.RAYS_QUAD, engine_mod.RAYS_SKIP):
        e = make_engine(engine_mod, m, ang, n, keep_ray_steps=1, max_range_m=max_range, ray_kernel=rk)
        assert e.max_range_px == om.max_range_px
        e.set_particles(p, np.full(n, 1.0 / n))
        e.sensor_update(obs)
        got = e.ray_steps()
        assert np.array_equal(got, steps), (rk, int((got != steps).sum()))
        assert np.array_equal(e.log_weights(), logw), rk


@pytest.mark.parametrize("n,poll", [(3000, None), (3000, "0"), (9000, None)])
def test_update_graph_replay_is_bit_identical(orc, engine_mod, spielberg, n, poll, monkeypatch):
    """graph_mode 0 (the default) shortens small updates from the second one on: up to 8192 particles an update is three
    launches (scan -> table rows inside the resampling kernel, k_rays_skip on the static table, k_tiny_tail writing the
    result block to pinned memory, the host spinning on its stamp unless MCL_TINY_POLL=0); above that everything after
    the resampling kernel is replayed as one hipGraph.  graph_mode 1 is the launch-by-launch path.  Same seed on both:
    identical particles, weights, pose and resample indices over many updates, including sensor_update calls, a new scan
    every update and a change of the particle set in between."""
    if poll is not None:
        monkeypatch.setenv("MCL_TINY_POLL", poll)
    ang = orc.beam_angles(angle_step=18)                      # 61 beams: the stock configuration
    base = load("scan_Spielberg_map_origin.npz")["ranges"][::18].copy()
    rng = np.random.default_rng(8)
    scans = [np.clip(base + rng.normal(0, 0.05, base.size), 0.0, 30.0).astype(np.float32) for _ in range(12)]
    out = {}
    for gm in (1, 0):
        e = make_engine(engine_mod, spielberg, ang, n, seed=3, graph_mode=gm)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        rec = []
        for k, sc in enumerate(scans):
            if k == 5:
                e.sensor_update(sc)
            elif k == 8:
                e.init_particles_pose((0.1, 0.0, 0.05), n)   # graphs are rebuilt after the particle set was replaced
                e.update(ACTION, sc)
            else:
                e.update(ACTION, sc)
            rec.append((e.get_particles(), e.get_weights(), e.expected_pose(), e.resample_indices() if k != 5 else None))
        assert e.ray_kernel_name() == "k_rays_skip"
        out[gm] = rec
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert (a[3] is None and b[3] is None) or np.array_equal(a[3], b[3])
