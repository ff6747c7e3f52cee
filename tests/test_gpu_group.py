"""Several engines behind one handle (mcl_group_*): the reference's single process drives the
shards.  With one GPU on the test box the group runs TWO (or four) engines on device 0 -- the peer copies and peer
pointers then stay on one device, everything else (global CDF, children of a shard drawn from the global parent set,
parents fetched where they live, host-side maxima and sums) is the real path.  It must equal one engine holding all
particles bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_engine

pytestmark = pytest.mark.gpu
ACTION = (0.05, 0.0, 0.01)


def make_group(engine_mod, m, ang, n_per, shards, **cfg):
    g = engine_mod.Group([0] * shards, max_particles=n_per, **cfg)
    g.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    g.set_beam_angles(ang)
    return g


@pytest.mark.parametrize("shards", [2, 4])
@pytest.mark.parametrize("mode", ["multinomial", "systematic"])
def test_group_equals_one_engine(orc, engine_mod, spielberg, shards, mode):
    ang = orc.beam_angles(angle_step=18)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].copy()
    n = 8192
    rm = engine_mod.RESAMPLE_MULTINOMIAL if mode == "multinomial" else engine_mod.RESAMPLE_SYSTEMATIC
    one = make_engine(engine_mod, spielberg, ang, n, seed=7, resample_mode=rm)
    one.init_particles_pose((0.0, 0.0, 0.0), n)
    grp = make_group(engine_mod, spielberg, ang, n // shards, shards, seed=7, resample_mode=rm)
    grp.init_particles_pose((0.0, 0.0, 0.0), n)
    assert np.array_equal(grp.get_particles(), one.get_particles())
    for k in range(6):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        assert np.array_equal(grp.resample_indices(), one.resample_indices()), f"update {k}"
        assert np.array_equal(grp.get_particles(), one.get_particles()), f"update {k}"
        # normalised weights = w / sum(w): w is bit-identical, the sum is reduced per shard and then added (last-ulp order)
        np.testing.assert_allclose(grp.get_weights(), one.get_weights(), rtol=1e-13, atol=0, err_msg=f"update {k}")
        np.testing.assert_allclose(grp.expected_pose(), one.expected_pose(), rtol=0, atol=1e-12)
    xb = grp.exchange_bytes()
    # after the first update the shards exchange their compact lists (the particles that carry weight, 44 B each) ...
    assert xb["lists"] and xb["parent_records_from_peers"] == 0
    assert 0 < xb["weights_received_per_device"] <= (shards - 1) * (n // shards) * 44
    grp.close(); one.close()


def test_group_weights_exchange_when_a_shard_has_no_list(orc, engine_mod, spielberg, monkeypatch):
    """MCL_NO_COMPACT=1: no engine makes a list, so every update gathers the weights and reads the selected parents where they
    live -- the exchange of a first update, kept alive over several."""
    monkeypatch.setenv("MCL_NO_COMPACT", "1")
    ang = orc.beam_angles(angle_step=18)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].astype(np.float32).copy()
    n = 8192
    one = make_engine(engine_mod, spielberg, ang, n, seed=3)
    grp = make_group(engine_mod, spielberg, ang, n // 2, 2, seed=3)
    monkeypatch.delenv("MCL_NO_COMPACT")
    one.init_particles_pose((0.0, 0.0, 0.0), n)
    grp.init_particles_pose((0.0, 0.0, 0.0), n)
    for k in range(3):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        assert np.array_equal(grp.resample_indices(), one.resample_indices()), f"update {k}"
        assert np.array_equal(grp.get_particles(), one.get_particles()), f"update {k}"
    xb = grp.exchange_bytes()
    assert not xb["lists"] and xb["weights_received_per_device"] == (n // 2) * 8
    # only children whose parent lives in another shard fetch a record across: never more than all children
    assert 0 < xb["parent_records_from_peers"] <= n * 32
    grp.close(); one.close()


def test_group_with_the_sweep_kernel_and_set_particles(orc, engine_mod, spielberg):
    """Two shards of 65 536 particles x 1081 beams (k_rays_sweep in both) from host-supplied particles."""
    from conftest import tracking_cloud
    ang = orc.beam_angles(angle_step=1)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
    n = 131072
    p = tracking_cloud(np.random.default_rng(4), n)
    w = np.full(n, 1.0 / n)
    one = make_engine(engine_mod, spielberg, ang, n, seed=11)
    one.set_particles(p, w)
    grp = make_group(engine_mod, spielberg, ang, n // 2, 2, seed=11)
    grp.set_particles(p, w)
    for _ in range(3):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
    assert one.ray_kernel_name() == "k_rays_sweep"
    assert np.array_equal(grp.get_particles(), one.get_particles())
    np.testing.assert_allclose(grp.get_weights(), one.get_weights(), rtol=1e-13, atol=0)
    np.testing.assert_allclose(grp.expected_pose(), one.expected_pose(), rtol=0, atol=1e-12)
    grp.close(); one.close()


def test_group_redoes_the_ray_stage_when_a_fix_up_list_overflows(orc, engine_mod, spielberg):
    """debug_force_exact=2 sends every ray of k_rays_sweep to the fix-up lists, which overflow.  The group takes the maximum on
    the devices and waits only for the sums: it learns of the overflow there and runs the ray stage (now with the self-contained
    kernel) and the reductions once more -- same children and weights as one engine, whose update does the same by itself."""
    from conftest import tracking_cloud
    ang = orc.beam_angles(angle_step=3)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::3].astype(np.float32)
    n = 131072
    p = tracking_cloud(np.random.default_rng(5), n)
    w = np.full(n, 1.0 / n)
    one = make_engine(engine_mod, spielberg, ang, n, seed=13, debug_force_exact=2)
    one.set_particles(p, w)
    grp = make_group(engine_mod, spielberg, ang, n // 2, 2, seed=13, debug_force_exact=2)
    grp.set_particles(p, w)
    for k in range(3):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        assert np.array_equal(grp.resample_indices(), one.resample_indices()), f"update {k}"
    assert one.ray_kernel_name() == "k_rays_skip"                      # the fallback ran
    assert np.array_equal(grp.get_particles(), one.get_particles())
    np.testing.assert_allclose(grp.get_weights(), one.get_weights(), rtol=1e-13, atol=0)
    grp.close(); one.close()


def test_group_adaptive_resampling_equals_one_engine(orc, engine_mod, spielberg):
    """resample_neff_permille in a device group: the decision from the group's sums (sum w, sum w^2 over the shards),
    mcl_stage_keep on every device instead of the exchange.  Same kept / resampled pattern (both occur), parents, particles and
    weights as one engine with the option."""
    from conftest import tracking_cloud
    ang = orc.beam_angles(angle_step=9)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::9].copy()
    n = 3000
    p = tracking_cloud(np.random.default_rng(4), n, sig=(0.03, 0.03, 0.01))
    w = np.full(n, 1.0 / n)
    one = make_engine(engine_mod, spielberg, ang, n, seed=11, resample_neff_permille=20)
    one.set_particles(p, w)
    grp = make_group(engine_mod, spielberg, ang, n // 2, 2, seed=11, resample_neff_permille=20)
    grp.set_particles(p, w)
    kept = []
    for k in range(8):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        kept.append(not one.effective_sample_size()[1])
        assert np.array_equal(grp.resample_indices(), one.resample_indices()), f"update {k}"
        assert np.array_equal(grp.get_particles(), one.get_particles()), f"update {k}"
        np.testing.assert_allclose(grp.get_weights(), one.get_weights(), rtol=1e-13, atol=0, err_msg=f"update {k}")
        if kept[-1]:
            xb = grp.exchange_bytes()
            assert xb["weights_received_per_device"] == 0 and xb["parent_records_from_peers"] == 0     # nothing was exchanged
    assert any(kept) and not all(kept[1:]), kept
    grp.close(); one.close()


def test_group_set_particles_with_non_uniform_weights(orc, engine_mod, spielberg):
    """Host-supplied weights that differ between the shards (the second shard holds most of the mass): every shard is
    quantised against the maximum of the WHOLE set, so the global CDF -- and with it every child -- equals one engine's."""
    from conftest import tracking_cloud
    ang = orc.beam_angles(angle_step=18)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].astype(np.float32).copy()
    n = 8192
    rng = np.random.default_rng(8)
    p = tracking_cloud(rng, n)
    w = rng.random(n) * np.where(np.arange(n) < n // 2, 1e-3, 1.0)
    w /= w.sum()
    for mode in (engine_mod.RESAMPLE_MULTINOMIAL, engine_mod.RESAMPLE_SYSTEMATIC):
        one = make_engine(engine_mod, spielberg, ang, n, seed=5, resample_mode=mode)
        one.set_particles(p, w)
        grp = make_group(engine_mod, spielberg, ang, n // 2, 2, seed=5, resample_mode=mode)
        grp.set_particles(p, w)
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        idx = one.resample_indices()
        assert np.array_equal(grp.resample_indices(), idx)
        assert np.array_equal(idx, orc.eng_resample_indices(orc.eng_quantize_weights(w), mode, k53=orc.eng_philox_k53(5, 0, 0, n),
                                                            k0=orc.eng_philox_k0(5, 0)))
        assert (idx >= n // 2).mean() > 0.99                 # the children come from the heavy shard
        assert np.array_equal(grp.get_particles(), one.get_particles())
        grp.close(); one.close()


def test_group_with_a_shard_that_carries_no_weight(orc, engine_mod, spielberg):
    """All of the first shard's weights are zero: its compact list is empty (length 0, not "no list"), the merged CDF starts
    with a plateau, every child comes from the second shard -- and equals one engine's, also on the update after."""
    from conftest import tracking_cloud
    ang = orc.beam_angles(angle_step=18)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].astype(np.float32).copy()
    n = 20000
    rng = np.random.default_rng(18)
    p = tracking_cloud(rng, n)
    w = rng.random(n)
    w[: n // 2] = 0.0
    w[n // 2:][rng.random(n // 2) < 0.9] = 0.0            # a tenth of the second shard carries weight: that shard has a list
    w /= w.sum()
    one = make_engine(engine_mod, spielberg, ang, n, seed=5)
    one.set_particles(p, w)
    grp = make_group(engine_mod, spielberg, ang, n // 2, 2, seed=5)
    grp.set_particles(p, w)
    assert grp.engine(0).compact_list()[0] == 0 and grp.engine(1).compact_list()[0] == np.count_nonzero(w)      # an empty list and a short one
    for k in range(3):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        idx = one.resample_indices()
        assert np.array_equal(grp.resample_indices(), idx), f"update {k}"
        assert np.array_equal(grp.get_particles(), one.get_particles()), f"update {k}"
        if k == 0:
            assert (idx >= n // 2).all() and grp.exchange_bytes()["lists"]
    grp.close(); one.close()


def test_shard_cdf_follows_the_staged_weights(orc, engine_mod, spielberg):
    """After a staged (sharded) update the shard's own CDF must describe its NEW weights: mcl_sample_particles (the
    reference's visualize(), cpp:946-958) on a one-shard group equals the same call on a plain engine."""
    ang = orc.beam_angles(angle_step=18)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::18].copy()
    n = 4096
    one = make_engine(engine_mod, spielberg, ang, n, seed=3)
    one.init_particles_pose((0.0, 0.0, 0.0), n)
    grp = make_group(engine_mod, spielberg, ang, n, 1, seed=3)
    grp.init_particles_pose((0.0, 0.0, 0.0), n)
    u = np.random.default_rng(8).random(60)
    for _ in range(3):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        assert np.array_equal(grp.engine(0).sample_particles(60, u), one.sample_particles(60, u))
    grp.close(); one.close()


def test_group_argument_errors(engine_mod, spielberg, orc):
    with pytest.raises(engine_mod.EngineError):
        engine_mod.Group([], max_particles=16)
    with pytest.raises(engine_mod.EngineError):
        engine_mod.Group([0, 0], max_particles=16, weight_mode=engine_mod.WEIGHT_PRODUCT, keep_ray_steps=1)
    g = make_group(engine_mod, spielberg, orc.beam_angles(angle_step=60), 16, 2)
    with pytest.raises(engine_mod.EngineError):
        g.init_particles_pose((0, 0, 0), 31)            # not a multiple of the device count
    with pytest.raises(engine_mod.EngineError):
        g.update(ACTION, np.ones(19, np.float32))        # particles not set
    g.close()


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: the group's peer copies and cross-device events between DISTINCT devices")
@pytest.mark.parametrize("mode", ["multinomial", "systematic"])
def test_group_on_two_distinct_devices_equals_one_engine(orc, engine_mod, spielberg, mode):
    """The same comparison as test_group_equals_one_engine with the shards on devices 0 and 1: hipMemcpyPeerAsync between two
    devices, peer-pointer parent reads (first update) and the ev_ready / ev_children waits across devices.  Skipped on a
    one-GPU box -- there every group test puts its shards on device 0, which is a REHEARSAL of this path."""
    ang = orc.beam_angles(angle_step=4)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::4].astype(np.float32).copy()
    n = 262144
    rm = engine_mod.RESAMPLE_MULTINOMIAL if mode == "multinomial" else engine_mod.RESAMPLE_SYSTEMATIC
    one = make_engine(engine_mod, spielberg, ang, n, seed=7, resample_mode=rm)
    one.init_particles_pose((0.0, 0.0, 0.0), n)
    grp = engine_mod.Group([0, 1], max_particles=n // 2, seed=7, resample_mode=rm)
    grp.set_map(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    grp.set_beam_angles(ang)
    grp.init_particles_pose((0.0, 0.0, 0.0), n)
    for k in range(5):
        one.update(ACTION, obs)
        grp.update(ACTION, obs)
        assert np.array_equal(grp.resample_indices(), one.resample_indices()), f"update {k}"
        assert np.array_equal(grp.get_particles(), one.get_particles()), f"update {k}"
    assert grp.exchange_bytes()["lists"]
    grp.close(); one.close()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL with one rank per device")
@pytest.mark.parametrize("mode,native", [(0, "0"), (1, "0"), (0, "1"), (1, "1")])
def test_two_ranks_over_rccl_on_two_devices_equal_one_rank(tmp_path, mode, native):
    """ShardedFilter over the nccl (= RCCL) backend, one rank per DISTINCT device: the all-gather of the compact lists and the
    small all-reduces as bench.py --gpus 2 issues them (native "0": torch's collectives, the default; "1": the engine's own
    communicator, mcl_comm_update, MCL_DIST_NATIVE=1).  Skipped on a one-GPU box, where the same code runs over gloo."""
    from test_dist import run_world, check_equal
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    two = run_world("engine", d2, 2, 131072, 4, mode, True, MCL_TEST_BEAM_STEP="4", MCL_TEST_NCCL="1", MCL_DIST_NATIVE=native)
    assert int(two[0]["native"]) == int(native)
    one = run_world("engine", d1, 1, 262144, 4, mode, False, MCL_TEST_BEAM_STEP="4")
    check_equal(two, one, 131072)
    assert list(two[0]["kinds"]) == ["dense", "lists", "lists", "lists"]
