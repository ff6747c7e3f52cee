"""The C++ host mirror of the reference class surface (host/particle_filter_core.*) driven like
timer_update() drives the reference (cpp:761-778), compared with the reference's OWN known answers
(N=2000, angle_step=18, seed 42, init cloud at (0,0,0), one MCL) and with the
oracle chain over several updates.  With use_reference_draws the mirror consumes its std::mt19937 /
normal_distribution exactly like the reference does and injects the draws through the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def write_inputs(tmp_path, m, scan):
    mp, sp = tmp_path / "map.bin", tmp_path / "scan.bin"
    with open(mp, "wb") as f:
        f.write(f"{m.width} {m.height} {float(np.float32(m.resolution))!r} {m.origin_x!r} {m.origin_y!r}\n".encode())
        f.write(np.ascontiguousarray(m.data, np.int8).tobytes())
    np.asarray(scan, np.float32).tofile(sp)
    return str(mp), str(sp)


def run_demo(args):
    import __graft_entry__ as g
    g.build()
    out = subprocess.check_output([os.path.join(ROOT, "host", "mcl_demo"), *map(str, args)], timeout=300)
    return json.loads(out)


def test_host_mirror_reproduces_reference_known_answers(tmp_path, spielberg):
    scan = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"]
    mp, sp = write_inputs(tmp_path, spielberg, scan)
    r = run_demo([mp, sp, 2000, 18, 1, 42, 1])
    assert r["max_range_px"] == 207 and r["beams"] == 61
    assert r["init_p0"] == [-0.27511724721024677, 0.25771653484560064, 0.18954434226648892]      # Appendix B
    np.testing.assert_allclose(r["p0"], [-0.033223950649317921, 0.34465170716028698, -0.74871940806065618], rtol=1e-13)
    np.testing.assert_allclose(r["w0"], 1.0946110511321877e-14, rtol=1e-5)
    np.testing.assert_allclose(r["wmax"], 0.22817876493613812, rtol=1e-5)
    pose = r["poses"][0]
    want = [0.02491833902169947, 0.011643671276452626, 9.4908712839202129e-05]
    assert abs(pose[0] - want[0]) < 1e-6 and abs(pose[1] - want[1]) < 1e-6 and abs(pose[2] - want[2]) < 1e-6
    assert abs(r["sum_w"] - 1.0) < 1e-12 and r["viz_rows"] == 60 and r["updates"] == 1


def test_host_mirror_multi_update_vs_oracle(tmp_path, orc, spielberg, spielberg_oracle):
    scan = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"]
    mp, sp = write_inputs(tmp_path, spielberg, scan)
    N, K = 1500, 5
    r = run_demo([mp, sp, N, 9, K, 7, 1])
    a, obs = orc.beam_angles(angle_step=9), scan[::9].copy()
    s = orc.RefStream(7)
    p, w = orc.init_particles_pose(s, (0.0, 0.0, 0.0), N)
    T = orc.sensor_table(spielberg_oracle.max_range_px)
    for k in range(K):
        u, nrm = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
        o = orc.mcl_step(spielberg_oracle, p, w, (0.05, 0.0, 0.01), a, obs, T, u, nrm, want_steps=False)
        p, w = o["particles"], o["weights"]
        pose = orc.expected_pose(p, w)
        got = r["poses"][k]
        assert abs(got[0] - pose[0]) < 1e-5 and abs(got[1] - pose[1]) < 1e-5 and abs(got[2] - pose[2]) < 1e-5, (k, got, pose)
    np.testing.assert_allclose(r["p0"], p[:, 0], rtol=1e-9, atol=1e-9)


def test_host_mirror_native_philox_runs(tmp_path, spielberg):
    scan = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"]
    mp, sp = write_inputs(tmp_path, spielberg, scan)
    r = run_demo([mp, sp, 20000, 1, 5, 42, 0])          # 1081 beams: the log-domain weights stay finite
    assert r["beams"] == 1081 and abs(r["sum_w"] - 1.0) < 1e-10 and r["wmax"] > 0
    assert all(np.isfinite(v) for pose in r["poses"] for v in pose)
    assert abs(r["poses"][-1][0]) < 1.0 and abs(r["poses"][-1][1]) < 1.0


def test_one_rank_in_plain_cpp_through_the_engines_rccl_communicator(tmp_path, engine_mod, spielberg):
    """host/comm_demo: one rank of the one-process-per-GPU host in C++ against the C ABI -- no Python, no torch in the process, so
    the RCCL the engine finds (dlopen) is the ROCm installation's.  Rendezvous over a file, 4 x mcl_comm_update (the first one
    without lists: the dense exchange).  Poses and particles equal a plain engine's mcl_update (same seed: Philox keyed by the
    global index).  One GPU here, so one rank; with more GPUs the same binary is started once per rank
    (comm_demo ... <n_ranks> <rank> <id_file>)."""
    import __graft_entry__ as g
    g.build()
    scan = np.load(os.path.join(GOLDEN, "scan_Spielberg_map_origin.npz"))["ranges"]
    mp, sp = write_inputs(tmp_path, spielberg, scan)
    n, step, k, seed = 131072, 3, 4, 77
    dump = tmp_path / "particles.bin"
    out = subprocess.check_output([os.path.join(ROOT, "host", "comm_demo"), mp, sp, str(n), str(step), str(k), str(seed), "1", "0",
                                   str(tmp_path / "rccl_id")], timeout=300, env=dict(os.environ, MCL_DEMO_DUMP=str(dump)))
    r = json.loads(out.decode().strip().splitlines()[-1])        # (RCCL may print its version banner first)
    assert r["ranks"] == 1 and r["beams"] == 361 and r["last_exchange"] == "lists" and r["host_waits"] == 1
    angle_min, angle_inc = np.float32(-3.0 * np.pi / 4.0), np.float32((3.0 * np.pi / 2.0) / 1080.0)
    ang = (angle_min + np.arange(0, 1081, step, dtype=np.float32) * angle_inc).astype(np.float32)
    e = engine_mod.Engine(max_particles=n, seed=seed)
    e.set_map(spielberg.data, spielberg.resolution, spielberg.origin_x, spielberg.origin_y)
    e.set_beam_angles(ang)
    e.init_particles_pose((0.0, 0.0, 0.0), n)
    obs = scan[::step].astype(np.float32)
    for it in range(k):
        e.update((0.05, 0.0, 0.01), obs)
        np.testing.assert_allclose(r["poses"][it], e.expected_pose(), rtol=0, atol=1e-12)
    p = e.get_particles()
    assert r["p0"] == [p[0, 0], p[1, 0], p[2, 0]]
    assert np.array_equal(np.fromfile(dump, np.float64).reshape(3, n), p)        # every particle, bit for bit
    e.close()
