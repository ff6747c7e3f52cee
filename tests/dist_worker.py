"""Worker for the world_size>1 tests: one rank of a ShardedFilter over gloo (CPU, oracle-backed shard)
or over the real HIP engine (GPU box; ranks share the one GPU).  Writes its shard's result to an .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend_kind, out_dir, n_local, steps, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # MCL_TEST_NCCL=1: RCCL with one DISTINCT device per rank (needs at least `world` GPUs); else gloo, the ranks share GPU 0
    nccl = os.environ.get("MCL_TEST_NCCL") == "1"
    dev_index = rank if nccl else 0
    if nccl:
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from monte_carlo_localization_amd import maps
    from monte_carlo_localization_amd.dist import ShardedFilter
    from oracle import oracle as orc
    m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
    bstep = int(os.environ.get("MCL_TEST_BEAM_STEP", "30"))
    ang = orc.beam_angles(angle_step=bstep)
    obs = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"][::bstep].copy()
    rng = np.random.default_rng(123)
    ntot = n_local * world
    device_init = os.environ.get("MCL_TEST_DEVICE_INIT") == "1"        # the engine's own initialiser (Philox keyed by the global index)
    digest = os.environ.get("MCL_TEST_DIGEST") == "1"                  # checksums per 2^20 particles instead of the arrays
    sig = (0.03, 0.03, 0.01) if os.environ.get("MCL_TEST_TIGHT") == "1" else (0.5, 0.5, 0.4)     # tight: weights that stay comparable
    p = None if device_init else np.stack([rng.normal(0, sig[0], ntot), rng.normal(0, sig[1], ntot), rng.normal(0, sig[2], ntot)])
    mine = slice(rank * n_local, (rank + 1) * n_local)
    w = np.full(n_local, 1.0 / ntot)
    skewed = os.environ.get("MCL_TEST_SKEWED_WEIGHTS") == "1"          # host-supplied weights that differ between the shards
    if skewed:
        wall = rng.random(ntot) * np.where(np.arange(ntot) < ntot // 2, 1e-3, 1.0)
        wall /= wall.sum()
        w = wall[mine]
    if backend_kind == "oracle":
        from oracle_shard import OracleShard
        om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
        shard = OracleShard(om, ang, seed=2024, resample_mode=mode)
        shard.set_particles(p[:, mine], w)
        device = torch.device("cpu")
    else:
        from monte_carlo_localization_amd import engine
        shard = engine.Engine(max_particles=n_local, device=dev_index, seed=2024, resample_mode=mode,
                              debug_force_exact=int(os.environ.get("MCL_TEST_FORCE_EXACT", "0")),
                              resample_neff_permille=int(os.environ.get("MCL_TEST_NEFF", "0")))
        shard.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
        shard.set_beam_angles(ang)
        if device_init:
            shard.init_particles_pose((0.0, 0.0, 0.0), n_local, rank * n_local, ntot)
        else:
            shard.set_particles(p[:, mine], w)
        device = torch.device("cuda", dev_index)
    overlap = len(sys.argv) > 6 and sys.argv[6] == "overlap"
    sf = ShardedFilter(shard, n_local, device, overlap=overlap)
    if skewed:
        sf.set_particles(p[:, mine], w)                 # every shard quantises against the maximum of the whole set
    poses, kinds, waits, kept, neff = [], [], [], [], []
    # MCL_TEST_EXPECT_FAIL=1 (the failure protocol, dist.py): an update may raise ShardedUpdateError -- on EVERY rank, from the same
    # update; the worker notes (update, whose failure, seconds), initialises the particle set again and goes on
    expect_fail = os.environ.get("MCL_TEST_EXPECT_FAIL") == "1"
    failures = []
    import time
    from monte_carlo_localization_amd.engine import ShardedUpdateError
    for it in range(steps):
        t0 = time.perf_counter()
        try:
            pose = sf.update((0.05, 0.0, 0.01), obs)
        except ShardedUpdateError as ex:
            if not expect_fail:
                raise
            failures.append((it, int(ex.local), time.perf_counter() - t0))
            print(f"[worker {rank}] update {it}: {ex}", file=sys.stderr, flush=True)
            if device_init:
                shard.init_particles_pose((0.0, 0.0, 0.0), n_local, rank * n_local, ntot)
                sf.reset()
            else:
                sf.set_particles(p[:, mine], w)
            continue
        poses.append(pose)
        kinds.append(sf.exchange_bytes["kind"])
        waits.append(sf.host_waits)                     # stream synchronisations of this update (1: the device-ordered flow)
        kept.append(int(getattr(sf, 'kept_last', False)))   # adaptive resampling kept the set in this update
        neff.append(sf.effective_sample_size()[0] if hasattr(sf, 'effective_sample_size') else 0.0)
    if backend_kind == "oracle":
        parts, q, idx = shard.p, shard.q, shard.idx
    else:
        parts, idx = shard.get_particles(), shard.resample_indices()
        qt = torch.empty(n_local, dtype=torch.int64, device=device)
        shard.export_state(0, 0, 0, qt.data_ptr())
        q = qt.cpu().numpy().view(np.uint64)
    if digest:
        from conftest import block_digests
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), particles=block_digests(parts), q=block_digests(q), idx=block_digests(idx),
                 poses=np.array(poses), kinds=np.array(kinds), waits=np.array(waits), kept=np.array(kept), neff=np.array(neff), native=np.array(int(getattr(sf, 'native', False))), failures=np.array(failures, dtype=np.float64).reshape(-1, 3))
    else:
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), particles=parts, q=q, idx=idx, poses=np.array(poses), kinds=np.array(kinds), waits=np.array(waits), kept=np.array(kept), neff=np.array(neff), native=np.array(int(getattr(sf, 'native', False))), failures=np.array(failures, dtype=np.float64).reshape(-1, 3))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
