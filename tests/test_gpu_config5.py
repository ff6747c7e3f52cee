"""BASELINE.json configs[4]: 33 554 432 particles on Spielberg_map, sharded -- here 2 x 16 777 216 on the one test GPU.
AT THE STATED WORKLOAD (1081 beams; the *_1081 tests at the end of this file): one engine holding all 33 554 432 particles is
checked against the ORACLE -- every resample index of two updates, sampled log-weights, sampled children -- and the two hosts
of the sharded engine must equal that engine bit for bit.
At 361 beams (three updates, multinomial and systematic; the particle-count-dependent code -- the global CDF of 2^25 entries,
global int32 parents, the bitmap over the global indices, the list exchange -- is what this size is for): both hosts of the
sharded engine -- one process per shard over torch.distributed (gloo here, RCCL in bench.py) and mcl_group_* in one process --
are compared with ONE engine holding all 33 554 432 particles, bit for bit, through checksums of 2^20-particle blocks
(conftest.block_digests): parents, children, fixed-point weights."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, block_digests, make_engine
from test_dist import run_world

pytestmark = pytest.mark.gpu
ACTION = (0.05, 0.0, 0.01)
N_TOTAL = 33_554_432
STEPS = 3
BEAM_STEP = 3                     # 361 beams: peaked enough for the list exchange (a list holds up to a quarter of the set)
SEED = 2024                       # tests/dist_worker.py's seed


@pytest.fixture(scope="module")
def one_engine(orc, engine_mod, spielberg):
    """mode -> digests of one engine's third update (cached: two tests per mode use it)."""
    cache = {}

    def get(mode):
        if mode not in cache:
            import torch
            ang = orc.beam_angles(angle_step=BEAM_STEP)
            obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::BEAM_STEP].astype(np.float32).copy()
            e = make_engine(engine_mod, spielberg, ang, N_TOTAL, seed=SEED, resample_mode=mode)
            e.init_particles_pose((0.0, 0.0, 0.0), N_TOTAL)
            poses = []
            for _ in range(STEPS):
                e.update(ACTION, obs)
                poses.append(e.expected_pose())
            assert e.ray_kernel_name() == "k_rays_sweep" and e.compact_list()[1]
            qt = torch.empty(N_TOTAL, dtype=torch.int64, device=torch.device("cuda", 0))
            e.export_state(0, 0, 0, qt.data_ptr())
            cache[mode] = dict(particles=block_digests(e.get_particles()), idx=block_digests(e.resample_indices()),
                               q=block_digests(qt.cpu().numpy().view(np.uint64)), poses=np.array(poses))
            del qt
            e.close()
        return cache[mode]
    return get


@pytest.mark.parametrize("mode", [0, 1])
def test_two_ranks_of_16m_equal_one_engine_of_32m(tmp_path, one_engine, mode):
    want = one_engine(mode)
    two = run_world("engine", tmp_path, 2, N_TOTAL // 2, STEPS, mode, True, MCL_TEST_BEAM_STEP=str(BEAM_STEP), MCL_TEST_DEVICE_INIT="1",
                    MCL_TEST_DIGEST="1")
    for k in ("idx", "particles", "q"):
        assert np.array_equal(np.concatenate([z[k] for z in two], axis=-1), want[k]), k
    np.testing.assert_allclose(two[0]["poses"], want["poses"], rtol=0, atol=1e-11)
    assert np.array_equal(two[0]["poses"], two[1]["poses"])
    assert list(two[0]["kinds"]) == ["dense", "lists", "lists"]


@pytest.mark.parametrize("mode", [0, 1])
def test_group_of_two_16m_shards_equals_one_engine_of_32m(orc, engine_mod, spielberg, one_engine, mode):
    want = one_engine(mode)
    ang = orc.beam_angles(angle_step=BEAM_STEP)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::BEAM_STEP].astype(np.float32).copy()
    g = engine_mod.Group([0, 0], max_particles=N_TOTAL // 2, seed=SEED, resample_mode=mode)
    g.set_map(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    g.set_beam_angles(ang)
    g.init_particles_pose((0.0, 0.0, 0.0), N_TOTAL)
    poses = []
    for _ in range(STEPS):
        g.update(ACTION, obs)
        poses.append(g.expected_pose())
    assert g.exchange_bytes()["lists"]
    assert np.array_equal(block_digests(g.resample_indices()), want["idx"])
    assert np.array_equal(block_digests(g.get_particles()), want["particles"])
    np.testing.assert_allclose(np.array(poses), want["poses"], rtol=0, atol=1e-11)
    g.close()


# ---- the stated workload: 33 554 432 particles x 1081 beams -------------------------------------------------------------------
STEPS_FULL = 2


@pytest.fixture(scope="module")
def one_engine_1081(orc, engine_mod, spielberg, spielberg_oracle):
    """ONE engine with all 33 554 432 particles x 1081 beams, two updates from the device-made sigma = 0.5 m cloud, checked
    against the oracle as tests/test_gpu_full_size.py does at 4M (cpp:656-665 resampling under the spec's Philox draws,
    cpp:586-650 + 545-579 log-weights): returns the digests the sharded runs must reproduce."""
    import torch
    om = spielberg_oracle
    ang = orc.beam_angles(angle_step=1)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
    n = N_TOTAL
    e = make_engine(engine_mod, spielberg, ang, n, seed=SEED, resample_mode=0)
    assert e.planned_ray_kernel(n)[0] == "k_rays_sweep"
    e.init_particles_pose((0.0, 0.0, 0.0), n)
    rng = np.random.default_rng(11)
    p0 = e.get_particles()
    # the device-made cloud is the spec's (Philox keyed by the global index): a slice in the middle and the last particles
    for first in (0, n // 2 + 12345, n - 4096):
        np.testing.assert_allclose(p0[:, first:first + 4096], orc.eng_init_pose(SEED, 0, (0.0, 0.0, 0.0), first, 4096),
                                   rtol=1e-13, atol=1e-13)                   # Box-Muller: device libm vs glibc
    L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
    oi = orc.obs_index(obs, om)
    poses = []

    # first update: uniform weights (every particle carries weight: full-CDF path), the spread cloud
    e.update(ACTION, obs)
    poses.append(e.expected_pose())
    assert e.ray_kernel_name() == "k_rays_sweep"
    idx1 = e.resample_indices()
    want1 = orc.eng_resample_indices(orc.eng_quantize_weights(np.full(n, 1.0 / n)), 0, k53=orc.eng_philox_k53(SEED, 0, 0, n))
    assert np.array_equal(idx1, want1), f"{np.count_nonzero(idx1 != want1)} of {n} parents differ from the oracle (update 1)"
    del want1
    parts1, lw1 = e.get_particles(), e.log_weights()
    pick = rng.choice(n, 4096, replace=False)
    sub = pick[:1024]
    nrm = np.concatenate([orc.eng_philox_normals(SEED, 0, int(i), 1) for i in sub])
    np.testing.assert_allclose(parts1[:, sub], orc.motion_model(p0[:, idx1[sub]], ACTION, nrm), rtol=1e-13, atol=1e-13)
    logw, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts1[:, pick]), ang, oi, L)
    assert np.array_equal(lw1[pick], logw)
    del p0, idx1, nrm

    # second update: the peaked weights of the first, drawn through the compact parent list
    _, q1, _ = orc.eng_weights_from_log(lw1)
    e.update(ACTION, obs)
    poses.append(e.expected_pose())
    assert e.compact_list()[1]
    idx2 = e.resample_indices()
    want2 = orc.eng_resample_indices(q1, 0, k53=orc.eng_philox_k53(SEED, 1, 0, n))
    assert np.array_equal(idx2, want2), f"{np.count_nonzero(idx2 != want2)} of {n} parents differ from the oracle (update 2)"
    del want2, q1
    parts2, lw2 = e.get_particles(), e.log_weights()
    pick2 = rng.choice(n, 4096, replace=False)
    sub2 = pick2[:1024]
    nrm2 = np.concatenate([orc.eng_philox_normals(SEED, 1, int(i), 1) for i in sub2])
    np.testing.assert_allclose(parts2[:, sub2], orc.motion_model(parts1[:, idx2[sub2]], ACTION, nrm2), rtol=1e-13, atol=1e-13)
    logw2, _, _ = orc.eng_log_weights(om, np.ascontiguousarray(parts2[:, pick2]), ang, oi, L)
    assert np.array_equal(lw2[pick2], logw2)
    # the weights the engine reports are the spec's: exp(logw - max) normalised (the oracle's deterministic exp)
    w_o, q_o, _ = orc.eng_weights_from_log(lw2)
    qt = torch.empty(n, dtype=torch.int64, device=torch.device("cuda", 0))
    e.export_state(0, 0, 0, qt.data_ptr())
    q_e = qt.cpu().numpy().view(np.uint64)
    assert np.array_equal(q_e, q_o)
    del qt
    out = dict(particles=block_digests(parts2), idx=block_digests(idx2), q=block_digests(q_e), poses=np.array(poses))
    e.close()
    return out


def test_32m_x_1081_one_engine_against_the_oracle(one_engine_1081):
    """The fixture holds the assertions (all 2 x 33 554 432 resample indices, 2 x 4096 log-weights, 2 x 1024 children, all
    33 554 432 fixed-point weights against the oracle); this test makes them a line of the report."""
    assert one_engine_1081["idx"].size == N_TOTAL >> 20
    assert np.isfinite(one_engine_1081["poses"]).all()


def test_two_ranks_of_16m_x_1081_equal_the_oracle_checked_engine(tmp_path, one_engine_1081):
    want = one_engine_1081
    two = run_world("engine", tmp_path, 2, N_TOTAL // 2, STEPS_FULL, 0, True, MCL_TEST_BEAM_STEP="1", MCL_TEST_DEVICE_INIT="1",
                    MCL_TEST_DIGEST="1")
    for k in ("idx", "particles", "q"):
        assert np.array_equal(np.concatenate([z[k] for z in two], axis=-1), want[k]), k
    np.testing.assert_allclose(two[0]["poses"], want["poses"], rtol=0, atol=1e-11)
    assert list(two[0]["kinds"]) == ["dense", "lists"]


def test_group_of_two_16m_x_1081_equals_the_oracle_checked_engine(orc, engine_mod, spielberg, one_engine_1081):
    want = one_engine_1081
    ang = orc.beam_angles(angle_step=1)
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
    g = engine_mod.Group([0, 0], max_particles=N_TOTAL // 2, seed=SEED, resample_mode=0)
    g.set_map(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    g.set_beam_angles(ang)
    g.init_particles_pose((0.0, 0.0, 0.0), N_TOTAL)
    poses = []
    for _ in range(STEPS_FULL):
        g.update(ACTION, obs)
        poses.append(g.expected_pose())
    assert g.exchange_bytes()["lists"]
    assert np.array_equal(block_digests(g.resample_indices()), want["idx"])
    assert np.array_equal(block_digests(g.get_particles()), want["particles"])
    np.testing.assert_allclose(np.array(poses), want["poses"], rtol=0, atol=1e-11)
    g.close()
