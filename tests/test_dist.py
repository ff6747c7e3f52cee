"""N>1 path: monte_carlo_localization_amd/dist.py driven with world_size 2 over gloo.

CPU test: the shard is the oracle-backed stand-in (tests/oracle_shard.py); the sharded result must be
bit-identical to the unsharded one (exact integer CDF, global Philox counters, exact log-weight sums).
GPU test: the same, with the real HIP engine in both ranks (sharing the single GPU of the test box)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(kind, out_dir, world, n_local, steps, mode, overlap=False, **extra_env):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), kind, str(out_dir),
                                       str(n_local), str(steps), str(mode), "overlap" if overlap else "sync"], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(os.path.join(out_dir, f"rank{r}.npz")) for r in range(world)]


def check_equal(two, one, n_local):
    cat = lambda k: np.concatenate([z[k] for z in two], axis=-1)
    assert np.array_equal(cat("idx"), one[0]["idx"])            # parents, global indexing
    assert np.array_equal(cat("particles"), one[0]["particles"])
    assert np.array_equal(cat("q"), one[0]["q"])                # fixed-point weights on the common scale
    np.testing.assert_allclose(two[0]["poses"], one[0]["poses"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(two[1]["poses"], two[0]["poses"], rtol=0, atol=0)   # every rank reports the same pose


@pytest.mark.parametrize("mode,overlap", [(0, False), (1, False), (0, True)])
def test_two_ranks_equal_one_rank_gloo_cpu(tmp_path, mode, overlap):
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    two = run_world("oracle", d2, 2, 96, 3, mode, overlap)
    one = run_world("oracle", d1, 1, 192, 3, mode)
    check_equal(two, one, 96)
    # the first update has no lists yet (uniform start, nothing reduced), the later ones exchange lists
    assert list(two[0]["kinds"]) == ["dense", "lists", "lists"]


def test_four_ranks_equal_one_rank_gloo_cpu(tmp_path):
    """Four shards: the merged list CDF carries three non-trivial offsets, the sums all-reduce twelve per-rank slots."""
    d4, d1 = tmp_path / "w4", tmp_path / "w1"
    d4.mkdir(); d1.mkdir()
    four = run_world("oracle", d4, 4, 48, 3, 0, True)
    one = run_world("oracle", d1, 1, 192, 3, 0)
    check_equal(four, one, 48)
    assert list(four[3]["kinds"]) == ["dense", "lists", "lists"]


def test_two_ranks_dense_exchange_only_gloo_cpu(tmp_path):
    """The dense exchange (weights gathered, distinct parents fetched) on every update: what runs when some shard has no
    compact list."""
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    two = run_world("oracle", d2, 2, 96, 3, 0, False, MCL_DIST_NO_LISTS="1")
    one = run_world("oracle", d1, 1, 192, 3, 0)
    check_equal(two, one, 96)
    assert list(two[0]["kinds"]) == ["dense"] * 3


@pytest.mark.parametrize("mode", [0, 1])
def test_two_ranks_host_weights_that_differ_between_shards_gloo_cpu(tmp_path, mode):
    """ShardedFilter.set_particles: every shard scales its fixed-point weights by the maximum of the WHOLE set, so a set whose
    mass sits in the second shard resamples exactly like the unsharded one."""
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    two = run_world("oracle", d2, 2, 96, 1, mode, False, MCL_TEST_SKEWED_WEIGHTS="1")
    one = run_world("oracle", d1, 1, 192, 1, mode, False, MCL_TEST_SKEWED_WEIGHTS="1")
    check_equal(two, one, 96)
    assert (np.concatenate([z["idx"] for z in two]) >= 96).mean() > 0.9      # the parents come from the heavy shard


def check_void_update(res, world, failing_rank, update_index, n_updates):
    """Every rank raised from the SAME update (the failing rank with its own error, the others with the peer's), quickly; the
    updates after the re-initialisation succeeded and every rank reports the same poses."""
    for r in range(world):
        f = res[r]["failures"]
        assert f.shape == (1, 3), (r, f)
        assert int(f[0, 0]) == update_index
        # (the failing rank reports its own failure; another rank normally reports the peer's -- unless what the failing rank
        #  contributed to an exchange made one of its own stages fail too: then it reports that, also a void update)
        assert int(f[0, 1]) == 1 or r != failing_rank
        assert f[0, 2] < 10.0                                          # seconds from the call to the exception
        assert res[r]["poses"].shape == (n_updates - 1, 3) and np.isfinite(res[r]["poses"]).all()
        assert np.array_equal(res[r]["poses"], res[0]["poses"])


@pytest.mark.parametrize("update,call,overlap,void", [(1, "stage_resample_indices", False, 0), (2, "stage_resample_compact", False, 1),
                                                      (2, "export_compact", True, 2), (2, "stage_rays", True, 1), (3, "stage_weights", False, 2),
                                                      (1, "stage_distinct_parents", True, 0)])
def test_a_failing_rank_voids_the_update_on_every_rank_gloo_cpu(tmp_path, update, call, overlap, void):
    """dist.py's failure protocol over the CPU stand-in: rank 1 fails before one engine call of one update (MCL_DIST_FAIL) --
    the dense exchange of the first update, the list exchange of the later ones, the ray stage, the weights stage, the export
    that prepares the next update's exchange.  It still enters every collective; both ranks raise ShardedUpdateError from that
    update within seconds; after set_particles the next updates run."""
    res = run_world("oracle", tmp_path, 2, 96, 4, 0, overlap, MCL_DIST_FAIL=f"1:{update}:{call}", MCL_TEST_EXPECT_FAIL="1")
    # (`void` = index of the update that raised.  With overlap, export_compact runs at the END of an update, for the next one's
    #  exchange: the failure is the NEXT update's -- that exchange is entered all the same, then the update is void)
    check_void_update(res, 2, 1, void, 4)


def test_a_failing_rank_among_four_gloo_cpu(tmp_path):
    res = run_world("oracle", tmp_path, 4, 48, 4, 0, True, MCL_DIST_FAIL="2:2:stage_resample", MCL_TEST_EXPECT_FAIL="1")
    check_void_update(res, 4, 2, 1, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("flow", ["ordered", "sync"])
def test_a_failing_rank_voids_the_update_on_every_rank_hip_engine(tmp_path, flow):
    """The same with the real engine in both ranks (gloo, the ranks share the GPU): the device-ordered flow (the error word is
    written on the device, behind the summed vector) and the stage-by-stage one."""
    env = dict(MCL_DIST_FAIL="1:2:stage_resample", MCL_TEST_EXPECT_FAIL="1", MCL_TEST_BEAM_STEP="6")
    if flow == "sync":
        env["MCL_DIST_SYNC"] = "1"
    res = run_world("engine", tmp_path, 2, 65536, 4, 0, True, **env)
    check_void_update(res, 2, 1, 1, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["resample", "rays", "weights"])
def test_native_rccl_update_soft_failure_one_rank(tmp_path, stage):
    """mcl_comm_update's failure protocol with the one-rank RCCL communicator (MCL_COMM_FAIL: the rank reports a failure
    before that stage of update 2): the call still issues its collectives, returns an error at once -- ShardedUpdateError,
    `.local` -- the communicator stays usable, and after the re-initialisation the next updates succeed."""
    res = run_world("engine", tmp_path, 1, 65536, 4, 0, False, MCL_DIST_NATIVE="1", MCL_TEST_NCCL="1", MCL_TEST_BEAM_STEP="6",
                    MCL_COMM_FAIL=f"0:2:{stage}", MCL_TEST_EXPECT_FAIL="1")
    assert int(res[0]["native"]) == 1
    check_void_update(res, 1, 0, 1, 4)


@pytest.mark.gpu
def test_native_rccl_update_bounded_wait_aborts_the_communicator(tmp_path):
    """A collective that does not finish within MCL_COMM_TIMEOUT_MS (here: the rank's own stream is kept busy for three times
    the bound before the last all-reduce, MCL_COMM_FAIL=0:2:stall -- what a peer that never arrives looks like from the
    host): the wait ends, the communicator is aborted, the call returns MCL_ERR_TIMEOUT, later calls say "create again"."""
    import ctypes as C
    from monte_carlo_localization_amd import engine, maps
    from oracle import oracle as orc
    os.environ["MCL_COMM_TIMEOUT_MS"] = "150"
    os.environ["MCL_COMM_FAIL"] = "0:2:stall"
    try:
        m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
        ang = orc.beam_angles(angle_step=6)
        obs = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"][::6].copy()
        e = engine.Engine(max_particles=65536, seed=5)
        ok, why = e.comm_available()
        if not ok:
            pytest.skip(f"no RCCL: {why}")
        e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
        e.set_beam_angles(ang)
        e.init_particles_pose((0.0, 0.0, 0.0), 65536, 0, 65536)
        e.comm_create(e.comm_unique_id(), 1, 0)
        e.comm_update((0.05, 0.0, 0.01), obs)
        import time
        t0 = time.perf_counter()
        with pytest.raises(engine.ShardedUpdateError) as ei:
            e.comm_update((0.05, 0.0, 0.01), obs)
        assert ei.value.status == engine.MCL_ERR_TIMEOUT and not ei.value.local
        assert time.perf_counter() - t0 < 10.0
        with pytest.raises(engine.ShardedUpdateError, match="create again"):
            e.comm_update((0.05, 0.0, 0.01), obs)
        # a new communicator, the particles initialised again: the sharded update works again
        e.comm_create(e.comm_unique_id(), 1, 0)
        e.init_particles_pose((0.0, 0.0, 0.0), 65536, 0, 65536)
        os.environ.pop("MCL_COMM_FAIL")
        pose = e.comm_update((0.05, 0.0, 0.01), obs)
        assert np.isfinite(pose).all()
        e.close()
    finally:
        os.environ.pop("MCL_COMM_TIMEOUT_MS", None)
        os.environ.pop("MCL_COMM_FAIL", None)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,overlap", [(0, False), (1, False), (0, True), (1, True)])
def test_two_ranks_equal_one_rank_hip_engine(tmp_path, mode, overlap):
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    two = run_world("engine", d2, 2, 4096, 4, mode, overlap)
    one = run_world("engine", d1, 1, 8192, 4, mode)
    check_equal(two, one, 4096)
    assert list(two[0]["kinds"]) == ["dense", "lists", "lists", "lists"]


@pytest.mark.gpu
def test_two_ranks_dense_exchange_and_skewed_weights_hip_engine(tmp_path):
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    two = run_world("engine", d2, 2, 4096, 3, 0, False, MCL_DIST_NO_LISTS="1", MCL_TEST_SKEWED_WEIGHTS="1")
    one = run_world("engine", d1, 1, 8192, 3, 0, False, MCL_TEST_SKEWED_WEIGHTS="1")
    check_equal(two, one, 4096)
    assert list(two[0]["kinds"]) == ["dense"] * 3


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_device_ordered_update_waits_once_and_equals_the_stage_by_stage_flow(tmp_path, overlap):
    """The default flow on a GPU (dist.py `_update_ordered`, the mcl_stage_*_async calls): the two small exchanges read and
    write device memory and the host waits ONCE per update; MCL_DIST_SYNC=1 is the stage-by-stage flow with a wait per value.
    Same kernels in the same order: identical bits -- here with k_rays_sweep (131 072 particles x 361 beams per rank) and,
    in the one-rank run, against a plain engine's mcl_update."""
    do, ds, d1 = tmp_path / "ordered", tmp_path / "sync", tmp_path / "one"
    do.mkdir(); ds.mkdir(); d1.mkdir()
    n = 131072
    two = run_world("engine", do, 2, n, 4, 0, overlap, MCL_TEST_BEAM_STEP="3")
    ref = run_world("engine", ds, 2, n, 4, 0, overlap, MCL_TEST_BEAM_STEP="3", MCL_DIST_SYNC="1")
    one = run_world("engine", d1, 1, 2 * n, 4, 0, False, MCL_TEST_BEAM_STEP="3")
    for r in range(2):
        for k in ("idx", "particles", "q", "poses"):
            assert np.array_equal(two[r][k], ref[r][k]), (r, k)
        assert list(two[r]["kinds"]) == ["dense", "lists", "lists", "lists"]
        assert list(two[r]["waits"][1:]) == [1, 1, 1]                   # (the first update has no lists yet: stage by stage)
        assert min(ref[r]["waits"][1:]) >= 4
    check_equal(two, one, n)
    assert list(one[0]["waits"][1:]) == [1, 1, 1]
    # ... and the plain engine, same seed and particles: the sharded flow is the single engine's update bit for bit
    import __graft_entry__ as g
    g.build()
    from conftest import GOLDEN
    from monte_carlo_localization_amd import engine, maps
    from oracle import oracle as orc
    m = maps.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz"))
    rng = np.random.default_rng(123)
    p = np.stack([rng.normal(0, 0.5, 2 * n), rng.normal(0, 0.5, 2 * n), rng.normal(0, 0.4, 2 * n)])
    e = engine.Engine(max_particles=2 * n, seed=2024, resample_mode=0)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(orc.beam_angles(angle_step=3))
    e.set_particles(p, np.full(2 * n, 1.0 / (2 * n)))
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::3].copy()
    for _ in range(4):
        e.update((0.05, 0.0, 0.01), obs)
    assert e.ray_kernel_name() == "k_rays_sweep"
    assert np.array_equal(e.get_particles(), one[0]["particles"])
    assert np.array_equal(e.resample_indices(), one[0]["idx"])
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["lists", "overflow", "dense"])
def test_native_rccl_update_one_rank_equals_the_torch_collectives(tmp_path, variant):
    """The engine's own RCCL communicator (mcl_comm_*: the whole sharded update in one native call, the collectives on the
    engine's stream) -- what `--backend nccl` runs with one rank per device.  One GPU here, so ONE rank: the communicator, the
    collectives and the single host wait are real, the transfers are not (tests/test_gpu_group.py holds the two-device
    test).  Against the same rank over torch's collectives (MCL_DIST_NATIVE=0).  "overflow": debug_force_exact=2, the redo after
    a fix-up list overflow; "dense": the exchange of an update without lists (whole weights and records) on every update."""
    dn, dt = tmp_path / "native", tmp_path / "torch"
    dn.mkdir(); dt.mkdir()
    n = 131072
    extra = dict(MCL_TEST_BEAM_STEP="3", MCL_TEST_NCCL="1")
    if variant == "overflow":
        extra["MCL_TEST_FORCE_EXACT"] = "2"
    nat = run_world("engine", dn, 1, n, 4, 0, False, MCL_DIST_NATIVE="1", **extra, **(dict(MCL_COMM_NO_LISTS="1") if variant == "dense" else {}))
    ref = run_world("engine", dt, 1, n, 4, 0, False, MCL_DIST_NATIVE="0", **extra, **(dict(MCL_DIST_NO_LISTS="1") if variant == "dense" else {}))
    assert int(nat[0]["native"]) == 1 and int(ref[0]["native"]) == 0
    for k in ("idx", "particles", "q", "poses"):
        assert np.array_equal(nat[0][k], ref[0][k]), k
    assert list(nat[0]["kinds"]) == (["dense"] * 4 if variant == "dense" else ["dense", "lists", "lists", "lists"])
    if variant == "overflow":
        assert min(nat[0]["waits"][1:]) > 1                              # the redo ran
    elif variant == "lists":
        assert list(nat[0]["waits"]) == [2, 1, 1, 1]                     # (a dense update waits once more, for the weight total)
    else:
        assert list(nat[0]["waits"]) == [2, 2, 2, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["sync", "ordered", "native"])
def test_adaptive_resampling_in_the_sharded_hosts_equals_one_engine(tmp_path, host):
    """Config.resample_neff_permille (E9: keep the set while N_eff of the WHOLE set stays above r / 1000 of it) through dist.py:
    the decision from the summed vector (sum w, sum w^2 of all shards), mcl_stage_keep instead of the exchange, the carried
    log-weights added in the ray stage, the carry committed at the end of the update.  Three hosts: stage by stage and
    device-ordered over gloo (two ranks on the one GPU), the engine's RCCL communicator (one rank).  Against a plain engine
    with the same option: same kept / resampled pattern (both occur), same parents, particles and weights."""
    d = tmp_path / host
    d.mkdir()
    n, steps, r = 3000, 8, 20
    extra = dict(MCL_TEST_BEAM_STEP="9", MCL_TEST_NEFF=str(r), MCL_TEST_TIGHT="1")
    if host == "native":
        world, n_local = 1, n
        got = run_world("engine", d, 1, n, steps, 0, False, MCL_TEST_NCCL="1", MCL_DIST_NATIVE="1", **extra)
        assert int(got[0]["native"]) == 1
    else:
        world, n_local = 2, n // 2
        got = run_world("engine", d, 2, n_local, steps, 0, host == "ordered", **extra, **(dict(MCL_DIST_SYNC="1") if host == "sync" else {}))
    import __graft_entry__ as g
    g.build()
    from conftest import GOLDEN
    from monte_carlo_localization_amd import engine, maps
    from oracle import oracle as orc
    m = maps.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz"))
    rng = np.random.default_rng(123)
    p = np.stack([rng.normal(0, 0.03, n), rng.normal(0, 0.03, n), rng.normal(0, 0.01, n)])
    e = engine.Engine(max_particles=n, seed=2024, resample_mode=0, resample_neff_permille=r)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(orc.beam_angles(angle_step=9))
    e.set_particles(p, np.full(n, 1.0 / n))
    obs = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"][::9].copy()
    kept, neff = [], []
    for _ in range(steps):
        e.update((0.05, 0.0, 0.01), obs)
        kept.append(int(not e.effective_sample_size()[1]))
        neff.append(e.effective_sample_size()[0])
    assert 0 < sum(kept) < steps - 1, kept                                # both branches, as in the single-engine oracle test
    for rk in range(world):
        assert list(got[rk]["kept"]) == kept, (rk, list(got[rk]["kept"]), kept)
        np.testing.assert_allclose(got[rk]["neff"], neff, rtol=1e-12)     # N_eff of the WHOLE set (sums reduced in another order)
    cat = lambda k: np.concatenate([z[k] for z in got], axis=-1)
    assert np.array_equal(cat("particles"), e.get_particles())
    assert np.array_equal(cat("idx"), e.resample_indices())
    np.testing.assert_allclose(got[0]["poses"][-1], e.expected_pose(), rtol=0, atol=1e-12)
    e.close()


@pytest.mark.gpu
def test_device_ordered_update_redoes_the_ray_stage_after_a_list_overflow(tmp_path):
    """debug_force_exact=2 sends every ray to the fix-up lists, which overflow: the device-ordered update learns that from the
    summed vector (its last element), and every rank runs the ray stage and the exchanges again stage by stage -- same result
    as the stage-by-stage flow, whose ray stage falls back by itself."""
    do, ds = tmp_path / "ordered", tmp_path / "sync"
    do.mkdir(); ds.mkdir()
    n = 65536
    two = run_world("engine", do, 2, n, 3, 0, False, MCL_TEST_BEAM_STEP="3", MCL_TEST_FORCE_EXACT="2")
    ref = run_world("engine", ds, 2, n, 3, 0, False, MCL_TEST_BEAM_STEP="3", MCL_TEST_FORCE_EXACT="2", MCL_DIST_SYNC="1")
    for r in range(2):
        for k in ("idx", "particles", "q", "poses"):
            assert np.array_equal(two[r][k], ref[r][k]), (r, k)
        assert min(two[r]["waits"][1:]) > 1                            # the redo ran


@pytest.mark.gpu
@pytest.mark.parametrize("n_children,n_total,span", [(5000, 40000, 40000), (70000, 1 << 20, 3000), (4096, 4096, 4096), (1000, 33, 33)])
def test_engine_distinct_parents_and_records_at(n_children, n_total, span):
    """mcl_stage_distinct_parents (bitmap + popcount prefix) against numpy.unique, and mcl_export_records_at against the
    particles themselves: what dist.py's exchange is built from."""
    import torch
    from conftest import GOLDEN
    from monte_carlo_localization_amd import engine, maps
    import __graft_entry__ as g
    g.build()
    rng = np.random.default_rng(n_children)
    dev = torch.device("cuda", 0)
    m = maps.load_npz(os.path.join(GOLDEN, "map_Spielberg_map.npz"))
    n = 6000
    e = engine.Engine(max_particles=n, seed=1)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(np.linspace(-2.0, 2.0, 31).astype(np.float32))
    p = rng.normal(0, 1.0, (3, n))
    e.set_particles(p, np.full(n, 1.0 / n))
    par = rng.integers(0, span, n_children).astype(np.int32)
    if span > 100:
        par[: n_children // 2] = par[0]                      # a heavy parent, as after a peaked update
        par[-1] = n_total - 1                                # the last bit of the last word
    d_par = torch.from_numpy(par).to(dev)
    uniq = torch.empty(n_children, dtype=torch.int64, device=dev)
    slot = torch.empty(n_children, dtype=torch.int32, device=dev)
    k = e.stage_distinct_parents(d_par.data_ptr(), n_children, n_total, uniq.data_ptr(), slot.data_ptr())
    want_u, want_inv = np.unique(par, return_inverse=True)
    assert k == want_u.size
    assert np.array_equal(uniq[:k].cpu().numpy(), want_u)
    assert np.array_equal(slot.cpu().numpy(), want_inv.astype(np.int32))
    idx = torch.from_numpy(rng.integers(0, n, 777)).to(dev)
    out = torch.empty((777, 4), dtype=torch.float64, device=dev)
    e.export_records_at(idx.data_ptr(), 777, out.data_ptr())
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :3], p[:, idx.cpu().numpy()].T)
    e.close()
