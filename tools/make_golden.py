#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/), whose reference-faithful half is
pinned to the reference's own outputs recorded in SURVEY.md Appendix B (tests/test_oracle_known_answers.py).
The random streams are libstdc++'s (oracle/refdraws.cpp), seeded like the survey's harness
(rng_.seed(42); normal_dist_.reset()).  Run from the repo root; deterministic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from monte_carlo_localization_amd import maps  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
ACTION = (0.05, 0.0, 0.01)


def omap(name):
    m = maps.load_npz(os.path.join(G, f"map_{name}.npz"))
    return m, orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)


def scan_from(om, pose, ang):
    n = ang.size
    r, s = orc.cast_many(om, np.full(n, pose[0]), np.full(n, pose[1]), pose[2] + ang.astype(np.float64))
    return r, s


def main():
    sp, osp = omap("Spielberg_map")
    sb, osb = omap("sibal1")
    ang = orc.beam_angles()

    # scans from the origin
    for name, om in (("Spielberg_map", osp), ("sibal1", osb)):
        r, s = scan_from(om, (0.0, 0.0, 0.0), ang)
        np.savez_compressed(os.path.join(G, f"scan_{name}_origin.npz"), ranges=r, steps=s.astype(np.int16), angles=ang)

    # G1: sensor tables (full fp64)
    for P in (207, 239):
        np.savez_compressed(os.path.join(G, f"g1_sensor_table_P{P}.npz"), table=orc.sensor_table(P))

    # G2: cast_ray on random + edge-case rays
    for name, m, om in (("Spielberg_map", sp, osp), ("sibal1", sb, osb)):
        rng = np.random.default_rng(2)
        n = 4096
        W, H, res = m.width, m.height, om.resolution
        x = om.origin_x + rng.uniform(-5 * res, (W + 5) * res, n)
        y = om.origin_y + rng.uniform(-5 * res, (H + 5) * res, n)
        th = rng.uniform(-np.pi, np.pi, n)
        # edge cases: within one cell of the lower/left edge, axis-aligned and 45-degree rays, inside walls
        x[:256] = om.origin_x + rng.uniform(-1.5 * res, 1.5 * res, 256)
        y[256:512] = om.origin_y + rng.uniform(-1.5 * res, 1.5 * res, 256)
        th[512:768] = rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2, np.pi / 4, -np.pi / 4, 3 * np.pi / 4], 256)
        oy_, ox_ = np.nonzero(m.data > 50)
        k = rng.integers(0, oy_.size, 256)
        x[768:1024] = om.origin_x + (ox_[k] + rng.uniform(0, 1, 256)) * res
        y[768:1024] = om.origin_y + (oy_[k] + rng.uniform(0, 1, 256)) * res
        fy_, fx_ = np.nonzero(m.data == 0)
        k = rng.integers(0, fy_.size, 2048)
        x[1024:3072] = om.origin_x + (fx_[k] + rng.uniform(0, 1, 2048)) * res
        y[1024:3072] = om.origin_y + (fy_[k] + rng.uniform(0, 1, 2048)) * res
        r, s = orc.cast_many(om, x, y, th)
        np.savez_compressed(os.path.join(G, f"g2_cast_ray_{name}.npz"), x=x, y=y, theta=th, ranges=r, steps=s.astype(np.int16))

    # G3: one full MCL step, N=512, B=61 and B=121
    T = orc.sensor_table(osp.max_range_px)
    full_r, _ = scan_from(osp, (0.0, 0.0, 0.0), ang)
    for step in (18, 9):
        a = orc.beam_angles(angle_step=step)
        obs = full_r[::step].copy()
        s = orc.RefStream(42)
        N = 512
        p, w = orc.init_particles_pose(s, (0.0, 0.0, 0.0), N)
        # a first update so that the second starts from non-uniform weights
        u0, n0 = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
        r0 = orc.mcl_step(osp, p, w, ACTION, a, obs, T, u0, n0)
        u1, n1 = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
        r1 = orc.mcl_step(osp, r0["particles"], r0["weights"], ACTION, a, obs, T, u1, n1)
        prop = r0["particles"][:, r1["idx"]]
        np.savez_compressed(
            os.path.join(G, f"g3_mcl_step_B{a.size}.npz"), angles=a, obs=obs, action=np.array(ACTION),
            particles_in=r0["particles"], weights_in=r0["weights"], uniforms=u1, normals=n1,
            idx=r1["idx"], gathered=prop, particles_out=r1["particles"], steps=r1["steps"].astype(np.uint8),
            raw_weights=r1["raw_weights"], weights_out=r1["weights"],
            pose=orc.expected_pose(r1["particles"], r1["weights"]))

    # G4: underflow witness, N=64, B=1081
    s = orc.RefStream(42)
    N = 64
    p, w = orc.init_particles_pose(s, (0.0, 0.0, 0.0), N)
    u, nrm = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
    r = orc.mcl_step(osp, p, w, ACTION, ang, full_r, T, u, nrm)
    L = orc.eng_log_table(T)
    logw, _, _ = orc.eng_log_weights(osp, r["particles"], ang, orc.obs_index(full_r, osp), L)
    np.savez_compressed(os.path.join(G, "g4_underflow_B1081.npz"), particles_in=p, weights_in=w, uniforms=u, normals=nrm,
                        particles_out=r["particles"], ref_weights=r["weights"], ref_raw_weights=r["raw_weights"],
                        eng_logw=logw, steps=r["steps"].astype(np.uint8))

    # G5: 30-step closed loop, N=2000, B=61, vehicle driving along the straight
    a = orc.beam_angles(angle_step=18)
    s = orc.RefStream(42)
    N = 2000
    pose = np.array([0.0, 0.0, 0.0])
    p, w = orc.init_particles_pose(s, pose, N)
    poses, truth = [], []
    for k in range(30):
        # true vehicle: straight/arc kinematics with the same action, no noise
        dt, v, om_ = orc.motion_scalars(ACTION)
        if abs(om_) < 1e-6:
            pose = pose + np.array([v * dt * np.cos(pose[2]), v * dt * np.sin(pose[2]), 0.0])
        else:
            R, d = v / om_, om_ * dt
            pose = np.array([pose[0] + R * (np.sin(pose[2] + d) - np.sin(pose[2])),
                             pose[1] - R * (np.cos(pose[2] + d) - np.cos(pose[2])), pose[2] + d])
        obs, _ = scan_from(osp, pose, a)
        u, nrm = s.uniforms(N), s.normals(3 * N).reshape(N, 3)
        r = orc.mcl_step(osp, p, w, ACTION, a, obs, T, u, nrm, want_steps=False)
        p, w = r["particles"], r["weights"]
        poses.append(orc.expected_pose(p, w))
        truth.append(pose.copy())
    np.savez_compressed(os.path.join(G, "g5_trajectory_N2000_B61.npz"), poses=np.array(poses), truth=np.array(truth),
                        final_particles=p, final_weights=w)
    print("golden fixtures written to", G)
    for f in sorted(os.listdir(G)):
        print(f"  {f}: {os.path.getsize(os.path.join(G, f))} B")


if __name__ == "__main__":
    main()
