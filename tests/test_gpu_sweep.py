"""k_rays_sweep (MCL_RAYS_SWEEP, the default ray kernel from 65 536 particles) against the CPU oracle WITHOUT
keep_ray_steps: that is the configuration in which its hand-written beam walk runs (step output, probe counting and
scans that wrap take the plain per-ray path, which tests/test_gpu_parity.py and test_gpu_first.py cover through the
`sweep` parameter).  Log-weights are exact fp64 sums of fp32 table entries, so the comparison is
bit for bit: a single wrong ray step, a stale partial sum, a ray counted twice by the fix-up list or an undecided ray
dropped shows up as a different double."""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, make_engine, tracking_cloud

pytestmark = pytest.mark.gpu


def oracle_logw(orc, om, p, ang, obs, inv_squash=None):
    T = orc.sensor_table(om.max_range_px)
    L = orc.eng_log_table(T) if inv_squash is None else orc.eng_log_table(T, inv_squash)
    logw, _, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L)
    return logw


def sweep_logw(engine_mod, m, ang, p, obs, **cfg):
    n = p.shape[1]
    e = make_engine(engine_mod, m, ang, n, ray_kernel=engine_mod.RAYS_SWEEP, **cfg)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(obs)
    assert e.ray_kernel_name() == "k_rays_sweep"
    out = e.log_weights(), e.counters()
    e.close()
    return out


def scan1081():
    return np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)


@pytest.mark.parametrize("force", [0, 2, 1])
def test_tracking_cloud_1081_beams(orc, engine_mod, spielberg, spielberg_oracle, force):
    """BASELINE config #1's shape (tracking cloud x 1081 beams).  force = 2 / 1 sends EVERY ray through the parked-ray
    overflow of the beam walk into the fix-up list (level 2 / level 3): nothing may be counted twice or lost."""
    ang = orc.beam_angles(angle_step=1)
    n = {0: 8192, 2: 2048, 1: 512}[force]
    # the forced cases count rays: keep every particle inside its window (off-window pairs go to k_rays_far, which traces
    # in fp64 and never uses the fix-up list)
    p = tracking_cloud(np.random.default_rng(3), n, sig=(0.5, 0.5, 0.4) if force == 0 else (0.2, 0.2, 0.4))
    got, c = sweep_logw(engine_mod, spielberg, ang, p, scan1081(), debug_force_exact=force)
    if force:
        assert c["off_window_particles"] == 0
    assert np.array_equal(got, oracle_logw(orc, spielberg_oracle, p, ang, scan1081()))
    if force == 0:
        assert 0 < c["level2_rays"] < n * ang.size // 100
    if force == 2:
        assert c["level2_rays"] == n * ang.size
    if force == 1:
        assert c["exact_fallback_rays"] == n * ang.size


def test_scattered_particles_and_off_window_pairs(orc, engine_mod, sibal1, sibal1_oracle):
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=5)
    rng = np.random.default_rng(11)
    n = 3000
    p = np.stack([om.origin_x + rng.uniform(-2, 20, n), om.origin_y + rng.uniform(-2, 11, n), rng.uniform(-np.pi, np.pi, n)])
    obs = np.full(ang.size, 3.0, np.float32)
    got, c = sweep_logw(engine_mod, sibal1, ang, p, obs)
    assert np.array_equal(got, oracle_logw(orc, om, p, ang, obs))
    assert c["off_window_particles"] > 0


def test_full_turn_scan_identical_particles_wedge_edges(orc, engine_mod, sibal1, sibal1_oracle):
    om = sibal1_oracle
    ang = np.linspace(-np.pi, np.pi - 0.004, 720).astype(np.float32)
    rng = np.random.default_rng(31)
    n = 3000
    p = np.stack([rng.uniform(-1.0, 1.0, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-np.pi, np.pi, n)])
    p[:, :1000] = np.array([[0.3], [0.1], [0.7]])                       # one hot bucket of the sort
    p[2, 1000:1100] = np.pi * rng.integers(-4, 5, 100) / 8.0             # headings exactly on wedge edges
    obs = rng.uniform(0.5, 8.0, ang.size).astype(np.float32)
    got, _ = sweep_logw(engine_mod, sibal1, ang, p, obs)
    assert np.array_equal(got, oracle_logw(orc, om, p, ang, obs))


@pytest.mark.parametrize("mapname", ["Spielberg_map", "sibal1", "icra_2_clean", "first_map"])
@pytest.mark.parametrize("max_range,n_beams_step", [(12.0, 7), (5.0, 13), (3.3, 31)])
def test_all_maps_and_ranges(orc, engine_mod, maps_mod, mapname, max_range, n_beams_step):
    m = maps_mod.load_npz(os.path.join(GOLDEN, f"map_{mapname}.npz"))
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y, max_range_m=max_range)
    from monte_carlo_localization_amd import synth
    ang = orc.beam_angles(angle_step=n_beams_step)
    rng = np.random.default_rng(zlib.crc32(f"sweep{mapname}{n_beams_step}".encode()))
    n = 2500
    p = synth.global_cloud(rng, m, n)
    p[:2, :100] += rng.normal(0, 3.0, (2, 100))           # some land in walls / outside the map
    p[:, 100:1100] = p[:, 100:101] + rng.normal(0, 0.05, (3, 1000))     # and a tight cluster: full 64-lane groups of near-identical rays
    obs = rng.uniform(0.0, max_range * 1.2, ang.size).astype(np.float32)
    got, _ = sweep_logw(engine_mod, m, ang, p, obs, max_range_m=max_range, squash_factor=3.1)
    assert np.array_equal(got, oracle_logw(orc, om, p, ang, obs, 1.0 / 3.1))


def test_cluster_near_map_corner_and_nonfinite(orc, engine_mod, sibal1, sibal1_oracle):
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=3)
    rng = np.random.default_rng(21)
    n = 2000
    p = np.stack([om.origin_x + rng.uniform(-0.2, 1.0, n), om.origin_y + rng.uniform(-0.2, 1.0, n), rng.uniform(-np.pi, np.pi, n)])
    bad = np.array([0, 1, 2, 3, 4])
    p[0, 0] = np.nan; p[1, 1] = np.inf; p[0, 2] = -np.inf; p[2, 3] = np.nan; p[0, 4] = 1e300
    obs = np.full(ang.size, 1.0, np.float32)
    got, _ = sweep_logw(engine_mod, sibal1, ang, p, obs)
    fin = np.setdiff1d(np.arange(n), bad)
    assert np.array_equal(got[fin], oracle_logw(orc, om, p[:, fin], ang, obs))


@pytest.mark.parametrize("g", [1, 2, 4, 8, 16])
def test_every_wedge_group_size_gives_the_same_sums(orc, engine_mod, spielberg, spielberg_oracle, g, monkeypatch):
    """MCL_SWEEP_G (read at mcl_create) sets how many wedges a work item walks before it takes the next run of units (one by
    default; every wedge's sum joins the slot accumulator atomically): the log-weights must not depend on it."""
    monkeypatch.setenv("MCL_SWEEP_G", str(g))
    ang = orc.beam_angles(angle_step=3)
    n = 70000                                                    # 69 units: runs of several units and single ones
    p = tracking_cloud(np.random.default_rng(5), n)
    obs = scan1081()[::3].copy()
    got, _ = sweep_logw(engine_mod, spielberg, ang, p, obs)
    assert np.array_equal(got, oracle_logw(orc, spielberg_oracle, p, ang, obs))


def test_default_kernel_at_size_and_updates_match_cell(orc, engine_mod, spielberg):
    """AUTO picks k_rays_sweep at 65 536 particles x 1081 beams; ten updates give the same particles, weights and pose as
    k_rays_cell, bit for bit."""
    ang = orc.beam_angles(angle_step=1)
    n = 65536
    out = {}
    for name, rk in (("auto", engine_mod.RAYS_AUTO), ("cell", engine_mod.RAYS_CELL)):
        e = make_engine(engine_mod, spielberg, ang, n, ray_kernel=rk, seed=9)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        for _ in range(10):
            e.update((0.05, 0.0, 0.01), scan1081())
        out[name] = (e.ray_kernel_name(), e.get_particles(), e.get_weights(), e.expected_pose())
        e.close()
    assert out["auto"][0] == "k_rays_sweep" and out["cell"][0] == "k_rays_cell"
    for k in (1, 2, 3):
        assert np.array_equal(out["auto"][k], out["cell"][k])


def test_work_list_overflow_falls_back_on_every_update(orc, engine_mod, spielberg):
    """AUTO at 7.9M rays with debug_force_exact=2: every ray is undecided, the fix-up lists overflow and EVERY update
    re-runs its ray stage with k_rays_skip (the kernel choice and the graph eligibility are pure functions of the
    configuration, so an overflow can never poison the next update).  Four updates equal a plain k_rays_skip engine."""
    ang = orc.beam_angles(angle_step=9)
    obs = scan1081()[::9].copy()
    n = 65536
    out = {}
    for name, cfg in (("auto", dict(ray_kernel=engine_mod.RAYS_AUTO, debug_force_exact=2)), ("skip", dict(ray_kernel=engine_mod.RAYS_SKIP))):
        e = make_engine(engine_mod, spielberg, ang, n, seed=21, **cfg)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        names = []
        for _ in range(4):
            e.update((0.05, 0.0, 0.01), obs)
            names.append(e.ray_kernel_name())
        out[name] = (names, e.get_particles(), e.get_weights(), e.expected_pose())
        e.close()
    assert out["auto"][0] == ["k_rays_skip"] * 4          # what finally produced each update's log-weights
    for k in (1, 2, 3):
        assert np.array_equal(out["auto"][k], out["skip"][k])


def test_range_beyond_the_lds_windows_takes_the_global_fields(orc, engine_mod, spielberg):
    """k_rays_sweep's LDS windows are 256 cells wide, k_rays_cell's 280: a range of more than 243 px leaves neither enough
    play (mcl_rays_sweep.h sweep_window_fits; mcl_set_map's qside).  AUTO then runs k_rays_sweep on the wedge fields in global
    memory (k_rays_sweep<.., GLOBAL>), an explicit MCL_RAYS_CELL is refused; the log-weights equal the oracle's."""
    res = 0.0485                                                     # 12 m / 0.0485 = 247 px
    om = orc.OracleMap(spielberg.data, res, spielberg.origin_x, spielberg.origin_y)
    assert 243 < om.max_range_px <= 255
    ang = orc.beam_angles(angle_step=8)                              # 136 beams: 65 536 x 136 > 2^23 rays
    n = 65536
    p = tracking_cloud(np.random.default_rng(12), n, sig=(0.2, 0.2, 0.3))
    obs = scan1081()[::8].copy()
    e = engine_mod.Engine(max_particles=n, seed=1)
    e.set_map(spielberg.data, res, spielberg.origin_x, spielberg.origin_y)
    e.set_beam_angles(ang)
    k, why = e.planned_ray_kernel()
    assert k == "k_rays_sweep" and "global memory" in why
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(obs)
    assert e.ray_kernel_name() == "k_rays_sweep"
    assert np.array_equal(e.log_weights(), oracle_logw(orc, om, p, ang, obs))
    v = e.ray_kernel_variant()
    assert v["hybrid"] or v["global_fields"]              # (an evenly spaced scan: the hybrid form, unless MCL_SWEEP_HYBRID=0)
    assert e.counters()["off_window_particles"] <= (16 if v["hybrid"] else 0)
    e.close()
    e = engine_mod.Engine(max_particles=n, seed=1, ray_kernel=engine_mod.RAYS_CELL)
    e.set_map(spielberg.data, res, spielberg.origin_x, spielberg.origin_y)
    e.set_beam_angles(ang)
    assert e.planned_ray_kernel()[0] is None
    e.set_particles(p, np.full(n, 1.0 / n))
    with pytest.raises(engine_mod.EngineError):
        e.sensor_update(obs)
    e.close()


def test_long_range_map_single_unit_runs_and_sparse_cloud(orc, engine_mod, sibal1, sibal1_oracle):
    """240-px range (0.05 m map, 12 m): 11 cells of play per window.  A cloud spread over many cells makes k_sweep_plan
    cut the runs down (and still leaves some particles to k_rays_far); a tight one keeps long runs.  Same sums as the
    oracle in both."""
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=8)
    obs = np.full(ang.size, 4.0, np.float32)
    rng = np.random.default_rng(77)
    n = 70000
    for spread in (0.05, 1.5):
        p = np.stack([rng.normal(0.3, spread, n), rng.normal(0.1, spread, n), rng.uniform(-np.pi, np.pi, n)])
        got, c = sweep_logw(engine_mod, sibal1, ang, p, obs)
        assert np.array_equal(got, oracle_logw(orc, om, p, ang, obs)), spread


def test_three_updates_each_checked_against_the_oracle(orc, engine_mod, spielberg, spielberg_oracle):
    """The sort keeps its histogram all-zero between updates by clearing the used tiles only (k_hist_clear); a residue
    would misplace particles in the sorted order of the NEXT update and leave their sums unwritten.  Three updates of
    k_rays_sweep, the log-weights of the particles each produced compared with the oracle's."""
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=8)
    base = scan1081()[::8].copy()
    rng = np.random.default_rng(6)
    n = 70000
    e = make_engine(engine_mod, spielberg, ang, n, ray_kernel=engine_mod.RAYS_SWEEP, seed=5)
    e.init_particles_pose((0.0, 0.0, 0.0), n)
    T = orc.sensor_table(om.max_range_px)
    L = orc.eng_log_table(T)
    for k in range(3):
        obs = np.clip(base + rng.normal(0, 0.03, base.size), 0.0, 30.0).astype(np.float32)
        e.update((0.05, 0.0, 0.01), obs)
        p = e.get_particles()
        logw, _, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L)
        assert e.ray_kernel_name() == "k_rays_sweep"
        assert np.array_equal(e.log_weights(), logw), k
    e.close()


@pytest.mark.parametrize("n_beams", [1024, 256])
def test_beam_count_multiple_of_256(orc, engine_mod, spielberg, spielberg_oracle, n_beams):
    """The beam walk requests the NEXT beam's direction on every trip, the last beam's included: with a beam count that is a
    multiple of 256 that entry used to lie past the end of the direction table (mcl_set_beam_angles now pads it by at
    least one entry).  65 536 particles through the default kernel choice, against the oracle."""
    n = 65536
    amin, amax = -3.0 * np.pi / 4.0, 3.0 * np.pi / 4.0
    ang = (np.float32(amin) + np.arange(n_beams, dtype=np.float32) * np.float32((amax - amin) / (n_beams - 1))).astype(np.float32)
    obs = np.interp(np.linspace(0, 1080, n_beams), np.arange(1081), scan1081()).astype(np.float32)
    p = tracking_cloud(np.random.default_rng(31), n, sig=(0.4, 0.4, 0.4))
    e = make_engine(engine_mod, spielberg, ang, n)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(obs)
    assert e.ray_kernel_name() == ("k_rays_sweep" if n * n_beams >= (8 << 20) else "k_rays_skip")
    got = e.log_weights()
    e.close()
    if n * n_beams < (8 << 20):           # AUTO stays on k_rays_skip at this ray count: force the walk
        got, _ = sweep_logw(engine_mod, spielberg, ang, p, obs)
    pick = np.random.default_rng(5).choice(n, 8192, replace=False)
    assert np.array_equal(got[pick], oracle_logw(orc, spielberg_oracle, p[:, pick], ang, obs))


@pytest.mark.parametrize("keep_steps", [0, 1])
def test_global_cloud_takes_the_windowed_far_pass(orc, engine_mod, sibal1, sibal1_oracle, spielberg, spielberg_oracle, keep_steps, monkeypatch):
    """(MCL_NO_BUCKET_CUTS: with the cloud ordered by buckets the window play can hold -- the default since, see
    test_narrow_play_sparse_cloud_is_cut_at_bucket_borders -- far fewer particles of this cloud are left for the far pass.)
    The uniform cloud of a global re-localisation (cpp:401-446) on a map whose 240-px range leaves a window 11 cells of play:
    no 32 x 32-cell tile of particles fits, every (particle, quadrant) pair is flagged and goes to the windowed far pass
    (k_rays_skip<.., FAR> over the ordered list of flagged slots: 568-cell nibble windows, only the flagged quadrants' beams).
    Log-weights (and ray steps) equal the oracle's, bit for bit.  On Spielberg_map (44 cells of play) the same kind of cloud
    needs no far pass at all since the units of a sparse set are cut at the tile borders."""
    from monte_carlo_localization_amd import synth
    monkeypatch.setenv("MCL_NO_BUCKET_CUTS", "1")
    ang = orc.beam_angles(angle_step=4)
    obs = np.full(ang.size, 3.0, np.float32)
    n = 100000
    p = synth.global_cloud(np.random.default_rng(5), sibal1, n)
    # (cpp:438-439 puts a particle exactly on the corner of its cell: every ray of it is boundary-ambiguous, the work list
    #  overflows and the stage is redone by k_rays_skip -- correct, tested elsewhere, but not the path under test: move off the corners)
    p[:2] += np.random.default_rng(55).uniform(0.1, 0.9, (2, n)) * float(np.float32(sibal1.resolution))
    p[:, :7] = np.array([[np.nan, 1e12, 0.0, -3.0, 5.0, 0.5, 0.0], [0.0, 0.0, np.inf, 2.0, -1.0, 0.5, 0.0],
                         [0.1, 0.2, 0.3, np.nan, np.inf, 1e9, 0.0]])          # garbage rows ride along in the list
    e = make_engine(engine_mod, sibal1, ang, n, ray_kernel=engine_mod.RAYS_SWEEP, keep_ray_steps=keep_steps)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(obs)
    got, c = e.log_weights(), e.counters()
    assert e.ray_kernel_name() == "k_rays_sweep"             # no work-list overflow, no fallback
    assert c["off_window_particles"] > n // 2                # far beyond kFarWindowedMin: the windowed pass took the list
    # (rows 0-4 are not finite: the reference's int cast of them is undefined behaviour, no common answer to compare with;
    #  row 5's huge-but-finite heading is marched literally, like the oracle does)
    pick = np.concatenate([np.arange(5, 16), 16 + np.random.default_rng(6).choice(n - 16, 6000, replace=False)])
    T = orc.sensor_table(sibal1_oracle.max_range_px)
    want, steps, _ = orc.eng_log_weights(sibal1_oracle, np.ascontiguousarray(p[:, pick]), ang, orc.obs_index(obs, sibal1_oracle),
                                         orc.eng_log_table(T), want_steps=True)
    assert np.array_equal(got[pick], want)
    if keep_steps:
        assert np.array_equal(e.ray_steps()[pick], steps)
    e.close()
    # Spielberg_map, same kind of cloud: units cut at the tile borders fit their windows (the default order again)
    monkeypatch.delenv("MCL_NO_BUCKET_CUTS")
    q = synth.global_cloud(np.random.default_rng(7), spielberg, 200000)
    q[:2] += np.random.default_rng(8).uniform(0.1, 0.9, (2, 200000)) * float(np.float32(spielberg.resolution))
    obs2 = scan1081()[::4].copy()
    got2, c2 = sweep_logw(engine_mod, spielberg, ang, q, obs2)
    assert c2["off_window_particles"] < 0.02 * 200000
    pick2 = np.random.default_rng(9).choice(200000, 4000, replace=False)
    assert np.array_equal(got2[pick2], oracle_logw(orc, spielberg_oracle, q[:, pick2], ang, obs2))
    # a handful of stragglers beside a tight cloud on the plain grid of units: below the threshold, k_rays_far keeps them
    # (with the cuts they are a unit of their own: test_far_apart_clusters_need_no_far_pass)
    monkeypatch.setenv("MCL_NO_BUCKET_CUTS", "1")
    t = tracking_cloud(np.random.default_rng(7), 70000, sig=(0.2, 0.2, 0.4))
    t[0, :500] += 40.0
    got3, c3 = sweep_logw(engine_mod, spielberg, ang, t, obs2)
    assert 0 < c3["off_window_particles"] < 2048
    assert np.array_equal(got3[:2000], oracle_logw(orc, spielberg_oracle, t[:, :2000], ang, obs2))


def test_2p5_cm_cells_range_of_479_px(orc, engine_mod, sibal1):
    """cpp:195 puts no bound on MAX_RANGE_PX = int(max_range / resolution): a 0.025 m map at 12 m is 479 px (the float32
    resolution is a hair above 0.025).  The step
    indices then need 16 bits and no LDS window holds a particle's reach, so AUTO -- at this small size -- runs k_rays_skip's global-field path
    (and MARCH the literal march): ray steps, log-weights and the children of a full update equal the oracle's."""
    grid = np.kron(sibal1.data, np.ones((2, 2), np.int8)).astype(np.int8)          # the same rooms at half the cell size
    res = 0.025
    om = orc.OracleMap(grid, res, sibal1.origin_x, sibal1.origin_y)
    P = om.max_range_px
    assert P == 479
    ang = orc.beam_angles(angle_step=12)
    rng = np.random.default_rng(41)
    n = 3000
    p = np.stack([rng.uniform(-7.0, 9.0, n), rng.uniform(-2.0, 6.0, n), rng.uniform(-np.pi, np.pi, n)])
    obs = (rng.uniform(0.3, 13.0, ang.size)).astype(np.float32)
    T = orc.sensor_table(om.max_range_px)
    L = orc.eng_log_table(T)
    want_logw, _, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L)
    _, want_steps = orc.cast_many(om, np.repeat(p[0], ang.size), np.repeat(p[1], ang.size),
                                  (p[2][:, None] + ang.astype(np.float64)[None, :]).ravel())
    for rk in (engine_mod.RAYS_AUTO, engine_mod.RAYS_MARCH):
        e = engine_mod.Engine(max_particles=n, seed=9, keep_ray_steps=1, ray_kernel=rk)
        e.set_map(grid, res, sibal1.origin_x, sibal1.origin_y)
        assert e.max_range_px == P
        e.set_beam_angles(ang)
        e.set_particles(p, np.full(n, 1.0 / n))
        e.sensor_update(obs)
        assert e.ray_kernel_name() == ("k_rays_skip" if rk == engine_mod.RAYS_AUTO else "k_rays_march")
        steps = e.ray_steps()
        assert steps.dtype == np.uint16 and steps.max() == P
        assert np.array_equal(steps.astype(np.int32).ravel(), np.asarray(want_steps, np.int32).ravel())
        assert np.array_equal(e.log_weights(), want_logw)
        # a full update through mcl_update: resampling from these weights, motion, rays again
        _, q, _ = orc.eng_weights_from_log(want_logw)
        e.update((0.05, 0.0, 0.01), obs)
        idx = e.resample_indices()
        assert np.array_equal(idx, orc.eng_resample_indices(q, 0, k53=orc.eng_philox_k53(9, 0, 0, n)))
        parts = e.get_particles()
        np.testing.assert_allclose(parts, orc.motion_model(p[:, idx], (0.05, 0.0, 0.01), orc.eng_philox_normals(9, 0, 0, n)), rtol=1e-13, atol=1e-13)
        logw2, _, _ = orc.eng_log_weights(om, parts, ang, orc.obs_index(obs, om), L)
        assert np.array_equal(e.log_weights(), logw2)
        e.close()
    # the LDS-windowed kernels cannot take this map: asking for one is refused (MCL_RAYS_SWEEP runs on the global wedge fields:
    # tests/test_gpu_sweep_global.py)
    e = engine_mod.Engine(max_particles=n, ray_kernel=engine_mod.RAYS_CELL)
    e.set_map(grid, res, sibal1.origin_x, sibal1.origin_y)
    e.set_beam_angles(ang)
    e.set_particles(p, np.full(n, 1.0 / n))
    with pytest.raises(engine_mod.EngineError):
        e.sensor_update(obs)
    e.close()


@pytest.mark.parametrize("how", ["hist", "radix"])
def test_both_orderings_give_the_oracles_sums(orc, engine_mod, spielberg, spielberg_oracle, monkeypatch, how):
    """The particles reach the ray kernel ordered by (tile, cell, heading) either through the counting sort with per-XCD
    histograms or -- from 3M particles by default -- through a radix sort of (key, index) pairs (MCL_SORT forces one).  The
    order decides which rays share a wave, never a result: the log-weights of a tracking cloud and of a uniform cloud (units
    cut at tile borders: the unit table then comes from the sorted keys / from the bucket offsets) equal the oracle's."""
    from monte_carlo_localization_amd import synth
    monkeypatch.setenv("MCL_SORT", how)
    ang = orc.beam_angles(angle_step=4)
    obs = scan1081()[::4].copy()
    n = 150000
    for cloud in ("tracking", "global"):
        p = tracking_cloud(np.random.default_rng(8), n) if cloud == "tracking" else synth.global_cloud(np.random.default_rng(9), spielberg, n)
        if cloud == "global":                # off the cell corners cpp:438-439 puts them on (all-ambiguous rays -> fallback to k_rays_skip)
            p[:2] += np.random.default_rng(99).uniform(0.1, 0.9, (2, n)) * float(np.float32(spielberg.resolution))
        got, _ = sweep_logw(engine_mod, spielberg, ang, p, obs)
        pick = np.random.default_rng(10).choice(n, 5000, replace=False)
        assert np.array_equal(got[pick], oracle_logw(orc, spielberg_oracle, p[:, pick], ang, obs)), cloud


@pytest.mark.parametrize("scan", ["regular-1081", "regular-1000", "regular-361", "jittered-721", "no-pad-env"])
def test_scan_edge_wedges_with_headings_apart(orc, engine_mod, spielberg, spielberg_oracle, monkeypatch, scan):
    """A lane whose scan begins or ends inside the wedge being walked has fewer beams there than its neighbours; it continues
    on VIRTUAL beams (the scan's angular grid continued beyond its ends, zero table columns) so that the lock-step walk covers
    every lane's real beams.  Uniform cloud: the 64 headings of a chunk are ~20 degrees apart, so chunks with such lanes are
    everywhere.  Scans whose wedge holds a whole number of beams (1081 over 270 degrees: 90), a fraction (1000: 83.3; 361: 30),
    an unevenly spaced scan (padding off: the grid cannot be continued) and MCL_NO_BEAM_PAD all give the oracle's sums."""
    from monte_carlo_localization_amd import synth
    rng = np.random.default_rng(21)
    if scan == "no-pad-env":
        monkeypatch.setenv("MCL_NO_BEAM_PAD", "1")
    if scan.startswith("regular") or scan == "no-pad-env":
        B = int(scan.split("-")[1]) if scan.startswith("regular") else 541
        ang = (np.float32(-0.75 * np.pi) + np.arange(B, dtype=np.float32) * np.float32(1.5 * np.pi / (B - 1))).astype(np.float32)
    else:
        B = 721
        base = -0.75 * np.pi + np.arange(B) * (1.5 * np.pi / (B - 1))
        ang = np.sort((base + rng.uniform(-0.45, 0.45, B) * (1.5 * np.pi / (B - 1))).astype(np.float32))
        assert np.all(np.diff(ang) > 0)
    full = scan1081()
    obs = np.interp(np.linspace(0.0, 1080.0, B), np.arange(1081), full).astype(np.float32)
    n = 120000
    p = synth.global_cloud(np.random.default_rng(22), spielberg, n)
    p[:2] += np.random.default_rng(23).uniform(0.1, 0.9, (2, n)) * float(np.float32(spielberg.resolution))
    got, _ = sweep_logw(engine_mod, spielberg, ang, p, obs)
    pick = np.random.default_rng(24).choice(n, 4000, replace=False)
    assert np.array_equal(got[pick], oracle_logw(orc, spielberg_oracle, p[:, pick], ang, obs))


@pytest.mark.parametrize("how", ["hist", "radix"])
def test_narrow_play_sparse_cloud_is_cut_at_bucket_borders(orc, engine_mod, sibal1, sibal1_oracle, monkeypatch, how):
    """sibal1's 239-px range leaves 12 cells of play in a 256-cell window: a uniform cloud ordered by whole 32 x 32 tiles lost most
    of its particles to the far pass.  A sparse set is now ordered by buckets the play can hold (8 x 8 cells here) and its units
    are cut at the bucket borders -- from the bucket offsets of the counting sort, or from where the radix-sorted keys change
    bucket -- so the windowed kernel keeps (nearly) all of them; the sums equal the oracle's either way, and over an update."""
    from monte_carlo_localization_amd import synth
    monkeypatch.setenv("MCL_SORT", how)
    ang = orc.beam_angles(angle_step=8)
    scan = scan1081()[::8].copy()
    n = 200000
    p = synth.global_cloud(np.random.default_rng(3), sibal1, n)
    p[:2] += np.random.default_rng(4).uniform(0.1, 0.9, (2, n)) * float(np.float32(sibal1.resolution))
    e = make_engine(engine_mod, sibal1, ang, n)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(scan)
    assert e.ray_kernel_name() == "k_rays_sweep"
    c = e.counters()
    assert c["off_window_particles"] < n // 50, c
    got = e.log_weights()
    pick = np.random.default_rng(5).choice(n, 4000, replace=False)
    assert np.array_equal(got[pick], oracle_logw(orc, sibal1_oracle, p[:, pick], ang, scan))
    e.update((0.05, 0.0, 0.01), scan)                      # resample + motion + the same stage on the children
    q = e.get_particles()
    assert np.array_equal(e.log_weights()[pick], oracle_logw(orc, sibal1_oracle, q[:, pick], ang, scan))
    e.close()


@pytest.mark.parametrize("how", ["hist", "radix"])
def test_far_apart_clusters_need_no_far_pass(orc, engine_mod, spielberg, spielberg_oracle, monkeypatch, how):
    """A set that sits on a few far-apart clusters (the steady state of a global re-localisation): the units of the sorted
    order are cut at the 32 x 32-cell tile borders, so no unit ends in one cluster and begins in the next (whose minority
    side no window would reach).  No particle is left for the far pass, whichever sort made the order; the oracle's sums."""
    monkeypatch.setenv("MCL_SORT", how)
    ang = orc.beam_angles(angle_step=4)
    obs = scan1081()[::4].copy()
    n = 180000
    rng = np.random.default_rng(31)
    centres = np.array([[0.0, 0.0, 0.0], [14.0, 3.0, 1.0], [-20.0, 9.0, -2.0], [5.5, -11.0, 3.0], [-7.0, 30.0, 0.5]])
    which = rng.integers(0, len(centres), n)
    p = (centres[which] + rng.normal(0.0, 1.0, (n, 3)) * np.array([0.12, 0.12, 0.2])).T.copy()
    got, c = sweep_logw(engine_mod, spielberg, ang, p, obs)
    assert c["off_window_particles"] == 0, c
    pick = rng.choice(n, 4000, replace=False)
    assert np.array_equal(got[pick], oracle_logw(orc, spielberg_oracle, p[:, pick], ang, obs))


def test_stage_timings_of_the_sweep_path_add_up(orc, engine_mod, spielberg):
    """The stage events of this path are bound to dispatches (stop events of hipExtLaunchKernelGGL: the resampling kernel, the
    kernel before the ray kernel, the ray kernel, k_combine_logw, k_scan_final) instead of recorded between kernels.  The six
    figures the host reads (hpp timing_stats_) must still be what they were: non-negative, the stages within the total, the ray
    kernel's own time within the ray stage."""
    ang = orc.beam_angles(angle_step=1)
    obs = scan1081()
    n = 131072
    e = make_engine(engine_mod, spielberg, ang, n, seed=3)
    e.set_particles(tracking_cloud(np.random.default_rng(1), n), np.full(n, 1.0 / n))
    for _ in range(3):
        e.update((0.05, 0.0, 0.01), obs)
        t = e.stage_timings()
        k = e.ray_kernel_ms()
        assert e.ray_kernel_name() == "k_rays_sweep"
        assert (t >= 0).all() and t[5] > 0
        assert 0 < k <= t[3] * 1.001                       # the ray kernel inside the ray stage (ordering + rays + fix-up)
        assert t[0] > 0 and t[4] > 0
        assert t[0] + t[2] + t[3] + t[4] <= t[5] * 1.02     # the device stages within the host's wall clock of the call
        assert t[0] + t[2] + t[3] + t[4] >= t[5] * 0.5
    e.close()
