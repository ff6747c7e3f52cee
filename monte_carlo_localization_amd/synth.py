"""Seeded synthetic inputs of SURVEY.md §8(d): Hokuyo-like 1081-beam geometry, a noise-free scan
ray-cast from a true pose (by the engine itself — same fixed-step semantics as cpp:611-650),
tracking-regime and global-regime particle clouds."""
from __future__ import annotations

import numpy as np


def beam_angles(n_beams: int = 1081, angle_step: int = 1) -> np.ndarray:
    """cpp:300-310 — angle_min + i*angle_increment evaluated in float32."""
    amin = np.float32(-3.0 * np.pi / 4.0)
    ainc = np.float32((3.0 * np.pi / 2.0) / 1080.0)
    i = np.arange(n_beams, dtype=np.float32)
    return (amin + i * ainc).astype(np.float32)[::angle_step].copy()


def scan_from_pose(eng, m, angles, pose) -> np.ndarray:
    """Noise-free scan: one particle at `pose`, its ray steps * resolution (float32), 12.0 on a miss.
    Uses a scratch engine with keep_ray_steps so the caller's engine state is untouched."""
    from . import engine as _e
    s = _e.Engine(max_particles=1, keep_ray_steps=1, max_range_m=eng.cfg.max_range_m)
    s.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    s.set_beam_angles(angles)
    s.set_particles(np.array(pose, np.float64).reshape(3, 1), np.ones(1))
    s.sensor_update(np.zeros(len(angles), np.float32))
    steps = s.ray_steps()[0].astype(np.int64)
    P = s.max_range_px
    res = float(np.float32(m.resolution))
    rng = np.where(steps >= P, np.float32(eng.cfg.max_range_m), (steps * res).astype(np.float32))
    s.close()
    return rng.astype(np.float32)


def tracking_cloud(rng, n, pose=(0.0, 0.0, 0.0), sig=(0.5, 0.5, 0.4)) -> np.ndarray:
    """cpp:392-397: N(pose, (0.5 m, 0.5 m, 0.4 rad)), theta wrapped."""
    p = np.empty((3, n))
    p[0] = pose[0] + rng.normal(0, sig[0], n)
    p[1] = pose[1] + rng.normal(0, sig[1], n)
    th = pose[2] + rng.normal(0, sig[2], n)
    p[2] = (th + np.pi) % (2 * np.pi) - np.pi
    return p


def global_cloud(rng, m, n) -> np.ndarray:
    """cpp:430-441: uniform over free cells, theta ~ U[0, 2pi)."""
    fy, fx = np.nonzero(m.data == 0)
    k = rng.integers(0, fy.size, n)
    res = float(np.float32(m.resolution))
    p = np.empty((3, n))
    p[0] = fx[k] * res + m.origin_x
    p[1] = fy[k] * res + m.origin_y
    p[2] = rng.uniform(0, 2 * np.pi, n)
    return p
