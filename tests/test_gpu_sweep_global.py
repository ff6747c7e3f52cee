"""k_rays_sweep<.., GLOBAL>: the beam walk of k_rays_sweep with its probes served from the wedge fields in GLOBAL memory
instead of a 256-cell LDS window -- what AUTO runs for MAX_RANGE_PX > 243 (cpp:195 puts no bound on it) and what
MCL_SWEEP_GLOBAL=1 (read at mcl_create) forces for any map.  Same reference rows (cpp:586-650 ray cast, 545-579 table
evaluation), same oracle, same bar: ray steps and log-weights bit for bit.

Part 1 re-runs the oracle comparisons of tests/test_gpu_sweep.py with the variant forced on the reference's own maps (ranges
of 57 .. 239 px): tracking and scattered clouds, wedge edges, every wedge-group size, non-finite and
off-map particles, forced level 2 / level 3, fix-up overflow, beam counts that are multiples of 256, scans that wrap, the sort's
orderings, several updates.
Part 2 is the long-range case proper: a 0.025 m map at 12 m (479 px, 16-bit step indices) and a 1000-px range (both in a ten-bit
cell field around an origin of the lane's own), through AUTO at size and through MCL_RAYS_SWEEP with step output."""
import os

import numpy as np
import pytest

import test_gpu_sweep as S
from conftest import GOLDEN, make_engine, tracking_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["global", "hybrid"])
def _global_fields(request, monkeypatch):
    """Both forms a range beyond the LDS window can take: the walk on the global wedge fields alone (MCL_SWEEP_HYBRID=0, round 4's),
    and the hybrid -- LDS windows as far as they reach, the global fields where a ray leaves its window (evenly spaced scans: the
    default).  With MCL_SWEEP_GLOBAL=1 the reference's own maps (57 .. 239 px) go through them too: there a ray leaves a window
    laid out for 211 px only now and then, at 479 px every long one does."""
    if "lds_windows" not in request.keywords:
        monkeypatch.setenv("MCL_SWEEP_GLOBAL", "1")
    # ("2": also for a set that was just loaded -- by default such a set, like the spread cloud of a re-localisation, takes the
    #  global-field form for its first update)
    monkeypatch.setenv("MCL_SWEEP_HYBRID", "2" if request.param == "hybrid" else "0")
    return request.param


# ---- part 1: the sweep tests whose assertions do not depend on where the probes are served from
test_tracking_cloud_1081_beams = S.test_tracking_cloud_1081_beams
test_full_turn_scan_identical_particles_wedge_edges = S.test_full_turn_scan_identical_particles_wedge_edges
test_all_maps_and_ranges = S.test_all_maps_and_ranges
test_cluster_near_map_corner_and_nonfinite = S.test_cluster_near_map_corner_and_nonfinite
test_every_wedge_group_size_gives_the_same_sums = S.test_every_wedge_group_size_gives_the_same_sums
test_work_list_overflow_falls_back_on_every_update = S.test_work_list_overflow_falls_back_on_every_update
test_three_updates_each_checked_against_the_oracle = S.test_three_updates_each_checked_against_the_oracle
test_beam_count_multiple_of_256 = S.test_beam_count_multiple_of_256
test_both_orderings_give_the_oracles_sums = S.test_both_orderings_give_the_oracles_sums
test_scan_edge_wedges_with_headings_apart = S.test_scan_edge_wedges_with_headings_apart
test_long_range_map_single_unit_runs_and_sparse_cloud = S.test_long_range_map_single_unit_runs_and_sparse_cloud


def test_scattered_particles_none_off_window(orc, engine_mod, sibal1, sibal1_oracle, _global_fields):
    """The scattered cloud of test_scattered_particles_and_off_window_pairs: every lane has an origin of its own, so only the
    particles OUTSIDE the padded grid are left to k_rays_far (the LDS windows lose a good part of this cloud)."""
    om = sibal1_oracle
    ang = orc.beam_angles(angle_step=5)
    rng = np.random.default_rng(11)
    n = 3000
    p = np.stack([om.origin_x + rng.uniform(-2, 20, n), om.origin_y + rng.uniform(-2, 11, n), rng.uniform(-np.pi, np.pi, n)])
    obs = np.full(ang.size, 3.0, np.float32)
    got, c = S.sweep_logw(engine_mod, sibal1, ang, p, obs)
    assert np.array_equal(got, S.oracle_logw(orc, om, p, ang, obs))
    px, py = (p[0] - om.origin_x) / float(sibal1.resolution), (p[1] - om.origin_y) / float(sibal1.resolution)
    outside = (px < -1.0) | (px >= sibal1.width) | (py < -1.0) | (py >= sibal1.height)
    if _global_fields == "global":
        assert 0 < c["off_window_particles"] <= int(outside.sum())
    else:                       # (the hybrid's windows lose a part of this cloud like the LDS form's: the far pass takes those)
        assert c["off_window_particles"] >= int(outside.sum()) > 0


def test_ray_steps_against_the_oracle(orc, engine_mod, spielberg, spielberg_oracle):
    """Step output takes the per-slot path of the kernel (MCL_SWG_TRIP in its second form): every step index."""
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=9)
    scan, _ = orc.cast_many(om, np.zeros(ang.size), np.zeros(ang.size), ang.astype(np.float64))
    n = 777
    p = tracking_cloud(np.random.default_rng(1), n)
    e = make_engine(engine_mod, spielberg, ang, n, ray_kernel=engine_mod.RAYS_SWEEP, keep_ray_steps=1, debug_count_probes=1)
    e.set_particles(p, np.full(n, 1.0 / n))
    e.sensor_update(scan)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    logw, steps, probes = orc.eng_log_weights(om, p, ang, orc.obs_index(scan, om), L, want_steps=True)
    assert np.array_equal(e.ray_steps(), steps)
    assert np.array_equal(e.log_weights(), logw)
    assert 0 < e.counters()["probes"] < probes
    e.close()


@pytest.mark.lds_windows
def test_global_fields_and_lds_windows_agree_over_updates(orc, engine_mod, spielberg, monkeypatch):
    """The two forms of the kernel side by side on one cloud over five updates: indices, particles, log-weights identical."""
    ang = orc.beam_angles(angle_step=4)
    obs = S.scan1081()[::4].copy()
    n = 131072
    engines = []
    for g in ("0", "1"):
        monkeypatch.setenv("MCL_SWEEP_GLOBAL", g)
        e = make_engine(engine_mod, spielberg, ang, n, seed=77)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        engines.append(e)
    assert "global memory" in engines[1].planned_ray_kernel()[1] and "global memory" not in engines[0].planned_ray_kernel()[1]
    for k in range(5):
        for e in engines:
            e.update((0.05, 0.0, 0.01), obs)
            assert e.ray_kernel_name() == "k_rays_sweep"
        assert np.array_equal(engines[0].resample_indices(), engines[1].resample_indices()), k
        assert np.array_equal(engines[0].log_weights(), engines[1].log_weights()), k
        assert np.array_equal(engines[0].get_particles(), engines[1].get_particles()), k
    for e in engines:
        e.close()


# ---- part 2: ranges no LDS window holds
def fine_sibal1(sibal1):
    return np.kron(sibal1.data, np.ones((2, 2), np.int8)).astype(np.int8)          # the same rooms at half the cell size


@pytest.mark.lds_windows
@pytest.mark.parametrize("res,want_P", [(0.025, 479), (0.012, 999)])
def test_long_ranges_steps_logw_and_a_full_update(orc, engine_mod, sibal1, res, want_P):
    """0.025 m cells at 12 m = 479 px and 0.012 m = 999 px: MCL_RAYS_SWEEP with step output --
    16-bit step indices, log-weights and the children of a full update equal the oracle's; particles in walls, outside the
    map and non-finite included."""
    grid = fine_sibal1(sibal1)
    om = orc.OracleMap(grid, res, sibal1.origin_x, sibal1.origin_y)
    P = om.max_range_px
    assert P == want_P
    ang = orc.beam_angles(angle_step=12)
    rng = np.random.default_rng(41)
    n = 3000 if res > 0.02 else 1200
    sx, sy = grid.shape[1] * res, grid.shape[0] * res
    p = np.stack([sibal1.origin_x + rng.uniform(-0.05 * sx, 1.05 * sx, n), sibal1.origin_y + rng.uniform(-0.05 * sy, 1.05 * sy, n),
                  rng.uniform(-np.pi, np.pi, n)])
    p[:, 100:700] = p[:, 100:101] + rng.normal(0, 0.02, (3, 600))                  # full 64-lane groups of near-identical rays
    obs = (rng.uniform(0.3, 13.0, ang.size)).astype(np.float32)
    L = orc.eng_log_table(orc.sensor_table(P))
    want_logw, _, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L)
    _, want_steps = orc.cast_many(om, np.repeat(p[0], ang.size), np.repeat(p[1], ang.size),
                                  (p[2][:, None] + ang.astype(np.float64)[None, :]).ravel())
    for keep in (1, 0):
        e = engine_mod.Engine(max_particles=n, seed=9, keep_ray_steps=keep, ray_kernel=engine_mod.RAYS_SWEEP)
        e.set_map(grid, res, sibal1.origin_x, sibal1.origin_y)
        e.set_beam_angles(ang)
        e.set_particles(p, np.full(n, 1.0 / n))
        e.sensor_update(obs)
        assert e.ray_kernel_name() == "k_rays_sweep"
        if keep:
            steps = e.ray_steps()
            assert steps.dtype == np.uint16 and 255 < steps.max() <= P
            assert np.array_equal(steps.astype(np.int32).ravel(), np.asarray(want_steps, np.int32).ravel())
        assert np.array_equal(e.log_weights(), want_logw)
        _, q, _ = orc.eng_weights_from_log(want_logw)
        e.update((0.05, 0.0, 0.01), obs)
        idx = e.resample_indices()
        assert np.array_equal(idx, orc.eng_resample_indices(q, 0, k53=orc.eng_philox_k53(9, 0, 0, n)))
        parts = e.get_particles()
        logw2, _, _ = orc.eng_log_weights(om, parts, ang, orc.obs_index(obs, om), L)
        assert np.array_equal(e.log_weights(), logw2)
        e.close()


@pytest.mark.lds_windows
def test_fine025_at_size_through_auto(orc, engine_mod, maps_mod, spielberg, _global_fields):
    """bench.py --map fine025 in small: 262 144 particles x 1081 beams on the 0.025 m map through AUTO (the global wedge
    fields), two updates; the log-weights of 1024 sampled particles of each equal the oracle's on the particles the engine
    produced, and every resample index of the second update equals the oracle's exact-CDF draw."""
    from monte_carlo_localization_amd import synth
    fine = maps_mod.synthetic_fine025(spielberg)
    om = orc.OracleMap(fine.data, fine.resolution, fine.origin_x, fine.origin_y)
    assert om.max_range_px == 479
    ang = synth.beam_angles()
    n, seed = 262144, 3
    e = make_engine(engine_mod, fine, ang, n, seed=seed)
    scan = synth.scan_from_pose(e, fine, ang, (0.0, 0.0, 0.0))
    e.set_particles(synth.tracking_cloud(np.random.default_rng(5), n), np.full(n, 1.0 / n))
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(scan, om)
    rng = np.random.default_rng(6)
    lw_prev = None
    for k in range(2):
        e.update((0.05, 0.0, 0.01), scan)
        assert e.ray_kernel_name() == "k_rays_sweep"
        if lw_prev is not None:
            _, q, _ = orc.eng_weights_from_log(lw_prev)
            assert np.array_equal(e.resample_indices(), orc.eng_resample_indices(q, 0, k53=orc.eng_philox_k53(seed, k, 0, n)))
        parts, lw = e.get_particles(), e.log_weights()
        pick = rng.choice(n, 1024, replace=False)
        logw, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts[:, pick]), ang, oi, L)
        assert np.array_equal(lw[pick], logw), k
        lw_prev = lw
    c = e.counters()
    v = e.ray_kernel_variant()
    assert v["hybrid"] == (_global_fields == "hybrid") and v["global_fields"] == (_global_fields == "global") and v["turned_directions"]
    # (global fields: every lane has its own origin; hybrid: the cloud's outliers -- 40 cells of play at 0.025 m are one metre -- go
    #  to the far pass)
    assert (c["off_window_particles"] == 0 if _global_fields == "global" else c["off_window_particles"] < n // 100)
    assert c["level2_rays"] < 0.02 * n * ang.size
    e.close()


@pytest.mark.lds_windows
@pytest.mark.parametrize("shape", [(400, 60), (60, 400)])
def test_narrow_map_long_range_particles_on_the_border(orc, engine_mod, shape):
    """A map NARROWER than a ray is long (60 x 400 cells at 0.025 m: 479 px of range), a spread cloud with particles on the
    border cells, in walls and outside: every step index and log-weight against the oracle.  The global-field form addresses the
    allocation of the sixteen mirrored fields from its start with offsets >= 0; a walk that leaves the grid runs into its field's
    tail of stop rows (csrc/mcl_wedge.h: sweep_global_layout; tests/test_sweep_addressing.py is the arithmetic).
    cpp:195 puts no bound on MAX_RANGE_PX, cpp:632-636 none on the map size."""
    H, W = shape
    rng = np.random.default_rng(8)
    grid = np.zeros((H, W), np.int8)
    grid[rng.random((H, W)) < 0.01] = 100                                  # a few pillars
    grid[0, :] = grid[-1, :] = 100                                          # walls along two sides, the other two open to the outside
    res, ox, oy = 0.025, -1.0, -2.0
    om = orc.OracleMap(grid, res, ox, oy)
    P = om.max_range_px
    assert P == 479 and min(H, W) < P
    ang = orc.beam_angles(angle_step=9)
    n = 2048
    p = np.stack([ox + rng.uniform(-0.02 * W * res, 1.02 * W * res, n), oy + rng.uniform(-0.02 * H * res, 1.02 * H * res, n),
                  rng.uniform(-np.pi, np.pi, n)])
    # the border cells themselves, both ends of both axes, headings along the axes and the diagonals
    edge = np.array([[ox + 0.5 * res, oy + 0.5 * res], [ox + (W - 0.5) * res, oy + 0.5 * res], [ox + 0.5 * res, oy + (H - 0.5) * res],
                     [ox + (W - 0.5) * res, oy + (H - 0.5) * res], [ox + 1e-9, oy + 0.3 * H * res], [ox + W * res - 1e-9, oy + 0.6 * H * res]])
    k = 0
    for ex, ey in edge:
        for th in np.arange(8) * (np.pi / 4):
            p[:, k] = (ex, ey, th)
            k += 1
    p[:, 200:328] = p[:, 200:201] + rng.normal(0, 0.01, (3, 128))           # two full waves of near-identical rays
    obs = rng.uniform(0.2, 13.0, ang.size).astype(np.float32)
    L = orc.eng_log_table(orc.sensor_table(P))
    want_logw, _, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(obs, om), L)
    _, want_steps = orc.cast_many(om, np.repeat(p[0], ang.size), np.repeat(p[1], ang.size),
                                  (p[2][:, None] + ang.astype(np.float64)[None, :]).ravel())
    for keep in (1, 0):
        e = engine_mod.Engine(max_particles=n, seed=9, keep_ray_steps=keep, ray_kernel=engine_mod.RAYS_SWEEP)
        e.set_map(grid, res, ox, oy)
        e.set_beam_angles(ang)
        e.set_particles(p, np.full(n, 1.0 / n))
        e.sensor_update(obs)
        assert e.ray_kernel_name() == "k_rays_sweep"
        if keep:
            assert np.array_equal(e.ray_steps().astype(np.int32).ravel(), np.asarray(want_steps, np.int32).ravel())
        assert np.array_equal(e.log_weights(), want_logw)
        e.close()


@pytest.mark.lds_windows
@pytest.mark.parametrize("glob", ["0", "1"])
def test_turned_and_fetched_beam_directions_agree(orc, engine_mod, spielberg, monkeypatch, glob):
    """The two ways the walk gets a beam's direction (mcl_rays_sweep.h: REC turns the grid direction by the scan increment and adds
    the beam's own offset; TAB fetches (cos, sin) per ray -- MCL_SWEEP_NO_REC=1, and what a scan that is not evenly spaced gets):
    same parents, particles and log-weights over five updates, in both forms of the kernel; the REC engine against the oracle."""
    monkeypatch.setenv("MCL_SWEEP_GLOBAL", glob)
    ang = orc.beam_angles(angle_step=2)
    obs = S.scan1081()[::2].copy()
    n = 131072
    engines = []
    for norec in (False, True):
        if norec:
            monkeypatch.setenv("MCL_SWEEP_NO_REC", "1")
        e = make_engine(engine_mod, spielberg, ang, n, seed=78)
        e.init_particles_pose((0.0, 0.0, 0.0), n)
        engines.append(e)
    for k in range(5):
        for e in engines:
            e.update((0.05, 0.0, 0.01), obs)
            assert e.ray_kernel_name() == "k_rays_sweep"
        assert np.array_equal(engines[0].resample_indices(), engines[1].resample_indices()), k
        assert np.array_equal(engines[0].log_weights(), engines[1].log_weights()), k
        assert np.array_equal(engines[0].get_particles(), engines[1].get_particles()), k
    om = orc.OracleMap(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    parts = engines[0].get_particles()
    pick = np.random.default_rng(3).choice(n, 2048, replace=False)
    logw, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts[:, pick]), ang, orc.obs_index(obs, om), orc.eng_log_table(orc.sensor_table(om.max_range_px)))
    assert np.array_equal(engines[0].log_weights()[pick], logw)
    for e in engines:
        e.close()


@pytest.mark.lds_windows
def test_relocalisation_on_a_long_range_map_does_not_alternate_between_the_forms(orc, engine_mod, maps_mod, spielberg, _global_fields):
    """A global re-localisation on the 479-px map through AUTO (default settings: this test overrides the module's MCL_SWEEP_HYBRID).
    The freshly initialised set takes the global-field form; a hybrid update that leaves many particles to the far pass (the cloud has
    not converged) is followed by an update in the global-field form -- a back-off, so the two cannot alternate forever --; every
    update's log-weights (sampled) equal the oracle's on the particles the engine produced."""
    if _global_fields != "hybrid":
        pytest.skip("one run: the default decision")
    import os as _os
    from monte_carlo_localization_amd import synth
    _os.environ.pop("MCL_SWEEP_HYBRID", None)
    fine = maps_mod.synthetic_fine025(spielberg)
    om = orc.OracleMap(fine.data, fine.resolution, fine.origin_x, fine.origin_y)
    ang = synth.beam_angles()
    n = 262144
    e = make_engine(engine_mod, fine, ang, n, seed=11)
    scan = synth.scan_from_pose(e, fine, ang, (0.0, 0.0, 0.0))
    e.init_global(n)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(scan, om)
    rng = np.random.default_rng(4)
    hist = []
    for k in range(8):
        e.update((0.05, 0.0, 0.01), scan)
        v, c = e.ray_kernel_variant(), e.counters()
        hist.append(("hybrid" if v["hybrid"] else "global" if v["global_fields"] else "lds", c["off_window_particles"]))
        parts, lw = e.get_particles(), e.log_weights()
        pick = rng.choice(n, 512, replace=False)
        logw, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts[:, pick]), ang, oi, L)
        assert np.array_equal(lw[pick], logw), (k, hist)
    e.close()
    assert hist[0][0] == "global", hist                         # a freshly initialised set
    assert all(f in ("hybrid", "global") for f, _ in hist), hist
    for (f0, off0), (f1, _) in zip(hist, hist[1:]):
        if f0 == "hybrid" and off0 >= 1024:
            assert f1 == "global", hist                         # the back-off
    assert any(f == "hybrid" for f, _ in hist), hist            # ... and the hybrid is tried
