"""First end-to-end GPU parity checks through the C ABI (expanded in the other test_gpu_* files)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_engine, tracking_cloud

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", ["march", "skip", "skip_forced_exact", "skip_forced_level2", "quad", "quad_forced_exact",
                                    "quad_forced_level2", "cell", "cell_forced_exact", "cell_forced_level2",
                                    "w_sweep", "w_sweep_forced_exact", "w_sweep_forced_level2"])
def test_ray_steps_and_logw_match_oracle(orc, engine_mod, spielberg, spielberg_oracle, kernel):
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=9)       # 121 beams
    scan, _ = orc.cast_many(om, np.zeros(ang.size), np.zeros(ang.size), ang.astype(np.float64))
    rng = np.random.default_rng(1)
    N = 777
    # k_rays_sweep's 256-cell windows leave 44 cells of play on this map: a cloud that stays inside them keeps the ray
    # counts below exact (off-window pairs are k_rays_far's, which has no level 2)
    p = tracking_cloud(rng, N, sig=(0.3, 0.3, 0.4)) if kernel[0] == "w" else tracking_cloud(rng, N)
    cfg = dict(keep_ray_steps=1, debug_count_probes=1)
    cfg["ray_kernel"] = {"m": engine_mod.RAYS_MARCH, "s": engine_mod.RAYS_SKIP, "q": engine_mod.RAYS_QUAD, "c": engine_mod.RAYS_CELL,
                         "w": engine_mod.RAYS_SWEEP}[kernel[0]]
    if kernel.endswith("forced_exact"):
        cfg["debug_force_exact"] = 1
    if kernel.endswith("forced_level2"):
        cfg["debug_force_exact"] = 2
    e = make_engine(engine_mod, spielberg, ang, N, **cfg)
    e.set_particles(p, np.full(N, 1.0 / N))
    e.sensor_update(scan)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(scan, om)
    logw, steps, probes = orc.eng_log_weights(om, p, ang, oi, L, want_steps=True)
    got = e.ray_steps()
    assert np.array_equal(got, steps), f"{(got != steps).sum()} of {steps.size} ray steps differ"
    assert np.array_equal(e.log_weights(), logw)          # exact fp64 sums of fp32 entries
    c = e.counters()
    if kernel.endswith("forced_exact"):
        assert c["exact_fallback_rays"] == N * ang.size
    if kernel.endswith("forced_level2"):
        assert c["level2_rays"] == N * ang.size and c["exact_fallback_rays"] < N * ang.size // 1000
    if kernel in ("skip", "quad", "cell", "w_sweep"):
        # level 1 hands only a small fraction of rays to level 2, and level 2 almost none to level 3
        assert c["level2_rays"] < N * ang.size // 100 and c["exact_fallback_rays"] < N * ang.size // 10000
        assert 0 < c["probes"] < probes
    if kernel == "march":
        assert c["probes"] == probes


def test_full_step_matches_reference_chain(orc, engine_mod, spielberg, spielberg_oracle):
    import __graft_entry__ as g
    g.smoke()


def test_small_update_path_against_the_oracle(orc, engine_mod, spielberg, spielberg_oracle):
    """From the second update on a small update takes the three-launch path (table rows of the scan computed inside the
    resampling kernel, k_rays_skip reading the static table through them, result block written to pinned memory): ray
    steps and log-weights of the particles it produced equal the oracle's, and the scalars it reports equal what the
    log-weights imply."""
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=18)
    base = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].astype(np.float32)
    rng = np.random.default_rng(4)
    n = 1500
    e = make_engine(engine_mod, spielberg, ang, n, seed=11, keep_ray_steps=1)
    e.init_particles_pose((0.0, 0.0, 0.0), n)
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    for k in range(4):
        scan = np.clip(base + rng.normal(0, 0.05, base.size), 0.0, 30.0).astype(np.float32)
        e.update((0.05, 0.0, 0.01), scan)
        p = e.get_particles()
        logw, steps, _ = orc.eng_log_weights(om, p, ang, orc.obs_index(scan, om), L, want_steps=True)
        assert np.array_equal(e.ray_steps(), steps), k
        assert np.array_equal(e.log_weights(), logw), k
        w = e.get_weights()
        ref = np.exp(logw - logw.max())
        assert np.allclose(w * (ref.sum() / w.sum()), ref, rtol=1e-12, atol=0.0), k
        assert e.ray_kernel_name() == "k_rays_skip"
    e.close()


def test_planned_ray_kernel_is_known_before_the_first_update(orc, engine_mod, spielberg, maps_mod):
    """mcl_get_planned_ray_kernel: the kernel class an update WILL get (the pure function mcl_update consults) and the reason,
    before anything runs -- the fast windowed kernel needs >= 65536 particles and 2^23 rays, beam angles that increase over
    less than a turn and MAX_RANGE_PX <= 243; anything else takes k_rays_skip, and says so."""
    ang = orc.beam_angles(angle_step=1)
    e = engine_mod.Engine(max_particles=131072)
    with pytest.raises(engine_mod.EngineError):
        e.planned_ray_kernel()                       # no map, no beams: MCL_ERR_NOT_READY
    e.set_map(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    e.set_beam_angles(ang)
    assert e.planned_ray_kernel()[0] == "k_rays_sweep"          # default: max_particles
    k, why = e.planned_ray_kernel(4000)
    assert k == "k_rays_skip" and "fewer than 65536" in why
    # ... and it is what runs
    rng = np.random.default_rng(3)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
    e.set_particles(tracking_cloud(rng, 131072), np.full(131072, 1.0 / 131072))
    e.update((0.05, 0.0, 0.01), obs)
    assert e.ray_kernel_name() == e.planned_ray_kernel()[0] == "k_rays_sweep"
    # beam angles that do not increase: no contiguous beam range per direction wedge
    shuffled = ang.copy(); shuffled[[10, 20]] = shuffled[[20, 10]]
    e.set_beam_angles(shuffled)
    k, why = e.planned_ray_kernel()
    assert k == "k_rays_skip" and "monotone" in why
    e.close()
    # a 0.025 m map: MAX_RANGE_PX = 479 (cpp:195 has no bound), beyond the 256-cell LDS windows: the same kernel on the wedge
    # fields in global memory
    fine = maps_mod.synthetic_fine025(spielberg)
    e = engine_mod.Engine(max_particles=131072)
    e.set_map(fine.data, fine.resolution, fine.origin_x, fine.origin_y)
    e.set_beam_angles(ang)
    assert e.max_range_px == 479
    k, why = e.planned_ray_kernel()
    assert k == "k_rays_sweep" and "global memory" in why
    # a configured kernel that cannot run with this map: 0 / None up front, MCL_ERR_UNSUPPORTED from the update
    e2 = engine_mod.Engine(max_particles=1024, ray_kernel=engine_mod.RAYS_CELL)
    e2.set_map(fine.data, fine.resolution, fine.origin_x, fine.origin_y)
    e2.set_beam_angles(ang)
    k2, why2 = e2.planned_ray_kernel()
    assert k2 is None and "MAX_RANGE_PX" in why2
    e.close(); e2.close()
