"""First end-to-end GPU parity checks through the C ABI (expanded in the other test_gpu_* files)."""
import numpy as np
import pytest

from conftest import make_engine, tracking_cloud

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
