"""BASELINE.json configs[4] at its TOTAL: 33 554 432 particles on Spielberg_map, sharded -- here 2 x 16 777 216 on the one test
GPU (361 beams keep an update at tens of milliseconds; the particle-count-dependent code -- the global CDF of 2^25 entries,
global int32 parents, the bitmap over the global indices, the list exchange -- is what this size is for).  Both hosts of the
sharded engine: one process per shard over torch.distributed (gloo here, RCCL in bench.py) and mcl_group_* in one process.
Everything is compared with ONE engine holding all 33 554 432 particles, bit for bit, through checksums of 2^20-particle
blocks (conftest.block_digests): parents, children, fixed-point weights of three updates; multinomial and systematic."""
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
