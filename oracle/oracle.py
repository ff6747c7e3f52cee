"""ctypes front-end of the CPU oracle (oracle/libmcl_oracle.so).

TEST INFRASTRUCTURE ONLY — see the header of mcl_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmcl_oracle.so")

WEIGHT_FRAC_BITS = 36


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("mcl_oracle.c", "refdraws.cpp", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _Map(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("resolution", C.c_double), ("origin_x", C.c_double), ("origin_y", C.c_double),
                ("max_range_m", C.c_double), ("max_range_px", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_ref_cast_ray.restype = C.c_float
        _lib.orc_ref_cast_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        _lib.orc_max_range_px.restype = C.c_int32
        _lib.orc_max_range_px.argtypes = [C.c_double, C.c_float]
        _lib.orc_normalize_angle.restype = C.c_double
        _lib.orc_normalize_angle.argtypes = [C.c_double]
        _lib.orc_eng_det_exp.restype = C.c_double
        _lib.orc_eng_det_exp.argtypes = [C.c_double]
        _lib.orc_ref_normalize.restype = C.c_double
        _lib.orc_eng_weights_from_log.restype = C.c_double
        _lib.orc_eng_philox_k0.restype = C.c_uint32
        _lib.orc_eng_philox_k0.argtypes = [C.c_uint64, C.c_uint32]
        _lib.rd_create.restype = C.c_void_p
        _lib.rd_create.argtypes = [C.c_uint32]
        _lib.rd_destroy.argtypes = [C.c_void_p]
        _lib.orc_omp_threads.restype = C.c_int
        _lib.orc_omp_threads.argtypes = [C.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class OracleMap:
    """What cpp:190-195 keeps of the OccupancyGrid."""

    def __init__(self, grid: np.ndarray, resolution, origin_x: float, origin_y: float,
                 max_range_m: float = 12.0):
        self.grid = _c(grid, np.int8)
        self.height, self.width = self.grid.shape
        self.resolution_f32 = np.float32(resolution)
        self.resolution = float(self.resolution_f32)          # cpp:191 (float32 -> double)
        self.origin_x, self.origin_y = float(origin_x), float(origin_y)
        self.max_range_m = float(max_range_m)
        self.max_range_px = int(lib().orc_max_range_px(self.max_range_m, C.c_float(self.resolution_f32)))
        self._c = _Map(self.grid.ctypes.data, self.width, self.height, self.resolution,
                       self.origin_x, self.origin_y, self.max_range_m, self.max_range_px)

    @property
    def ref(self):
        return C.byref(self._c)


def omp_threads(n: int = 0) -> int:
    return lib().orc_omp_threads(n)


def sensor_table(P, z_hit=0.80, z_short=0.01, z_max=0.07, z_rand=0.12, sigma_hit=8.0):
    """(P+1)^2 doubles, Eigen column-major linear order: out[d*(P+1)+r]; returned as
    array T[d, r] (so T[d, r] == sensor_model_table_(r, d))."""
    out = np.empty((P + 1) * (P + 1), dtype=np.float64)
    lib().orc_ref_sensor_table(C.c_int32(P), C.c_double(z_hit), C.c_double(z_short), C.c_double(z_max),
                               C.c_double(z_rand), C.c_double(sigma_hit), _p(out))
    return out.reshape(P + 1, P + 1)


def cast_ray(m: OracleMap, x, y, angle):
    step = C.c_int32(0)
    r = lib().orc_ref_cast_ray(m.ref, float(x), float(y), float(angle), C.byref(step))
    return float(r), int(step.value)


def cast_many(m: OracleMap, x, y, angle, use_omp=False):
    x, y, angle = _c(x, np.float64), _c(y, np.float64), _c(angle, np.float64)
    n = x.size
    ranges = np.empty(n, np.float32)
    steps = np.empty(n, np.int32)
    lib().orc_ref_cast_many(m.ref, C.c_int64(n), _p(x), _p(y), _p(angle), _p(ranges), _p(steps), int(use_omp))
    return ranges, steps


def beam_angles(n_beams=1081, angle_min=None, angle_increment=None, angle_step=1):
    """cpp:300-310: laser_angles_[i] = angle_min + i*angle_increment, all in float."""
    if angle_min is None:
        angle_min = np.float32(-3.0 * np.pi / 4.0)
    if angle_increment is None:
        angle_increment = np.float32((3.0 * np.pi / 2.0) / 1080.0)
    i = np.arange(n_beams, dtype=np.float32)  # size_t -> float conversion is exact for < 2^24
    ang = (np.float32(angle_min) + i * np.float32(angle_increment)).astype(np.float32)
    return ang[::angle_step].copy()


def motion_model(p_colmajor, action, normals, disp=(0.05, 0.025, 0.25)):
    """p_colmajor: (3, N) array [x row, y row, theta row] == Eigen N x 3 column-major memory."""
    p = _c(p_colmajor, np.float64).copy()
    N = p.shape[1]
    a = _c(action, np.float64)
    nrm = _c(normals, np.float64)
    lib().orc_ref_motion_model(C.c_int64(N), _p(p), _p(a), _p(nrm), C.c_double(disp[0]),
                               C.c_double(disp[1]), C.c_double(disp[2]))
    return p


def motion_scalars(action):
    a = _c(action, np.float64)
    dt, v, w = C.c_double(), C.c_double(), C.c_double()
    lib().orc_ref_motion_scalars(_p(a), C.byref(dt), C.byref(v), C.byref(w))
    return dt.value, v.value, w.value


def obs_index(obs, m: OracleMap):
    obs = _c(obs, np.float32)
    out = np.empty(obs.size, np.int32)
    lib().orc_ref_obs_index(_p(obs), C.c_int32(obs.size), C.c_double(m.resolution), C.c_int32(m.max_range_px), _p(out))
    return out


def sensor_model(m: OracleMap, p_colmajor, angles, obs, table, inv_squash=1.0 / 2.2, use_omp=False,
                 want_steps=True):
    p = _c(p_colmajor, np.float64)
    N = p.shape[1]
    angles, obs = _c(angles, np.float32), _c(obs, np.float32)
    B = angles.size
    table = _c(table, np.float64)
    w = np.empty(N, np.float64)
    steps = np.empty(N * B, np.int32) if want_steps else None
    t = np.zeros(3)
    rc = lib().orc_ref_sensor_model(m.ref, C.c_int64(N), _p(p), C.c_int32(B), _p(angles), _p(obs), _p(table),
                                    C.c_double(inv_squash), _p(w), _p(steps), int(use_omp), _p(t))
    if rc:
        raise RuntimeError(f"orc_ref_sensor_model rc={rc}")
    return w, (steps.reshape(N, B) if want_steps else None), t


def resample_indices(weights, uniforms):
    w, u = _c(weights, np.float64), _c(uniforms, np.float64)
    idx = np.empty(w.size, np.int32)
    with np.errstate(all="ignore"):
        lib().orc_ref_resample_indices(C.c_int64(w.size), _p(w), _p(u), _p(idx))
    return idx


def expected_pose(p_colmajor, w):
    p, w = _c(p_colmajor, np.float64), _c(w, np.float64)
    out = np.empty(3)
    lib().orc_ref_expected_pose(C.c_int64(p.shape[1]), _p(p), _p(w), _p(out))
    return out


def mcl_step(m: OracleMap, particles, weights, action, angles, obs, table, uniforms, normals,
             inv_squash=1.0 / 2.2, disp=(0.05, 0.025, 0.25), use_omp=False, want_steps=True):
    """One reference MCL step (cpp:652-694).  Returns dict with new particles/weights etc."""
    p = _c(particles, np.float64).copy()
    w = _c(weights, np.float64).copy()
    N = p.shape[1]
    angles, obs = _c(angles, np.float32), _c(obs, np.float32)
    B = angles.size
    a = _c(action, np.float64)
    table = _c(table, np.float64)
    u, nrm = _c(uniforms, np.float64), _c(normals, np.float64)
    idx = np.empty(N, np.int32)
    steps = np.empty(N * B, np.int32) if want_steps else None
    raw = np.empty(N, np.float64)
    t = np.zeros(6)
    rc = lib().orc_ref_mcl_step(m.ref, C.c_int64(N), _p(p), _p(w), _p(a), C.c_int32(B), _p(angles), _p(obs),
                                _p(table), C.c_double(inv_squash), _p(u), _p(nrm), C.c_double(disp[0]),
                                C.c_double(disp[1]), C.c_double(disp[2]), _p(idx), _p(steps), _p(raw),
                                int(use_omp), _p(t))
    if rc:
        raise RuntimeError(f"orc_ref_mcl_step rc={rc}")
    return dict(particles=p, weights=w, idx=idx, steps=(steps.reshape(N, B) if want_steps else None),
                raw_weights=raw, timing_ms=t)


class RefStream:
    """The reference's rng_/normal_dist_ pair (hpp:165-167) replayed with libstdc++."""

    def __init__(self, seed=42):
        self._h = lib().rd_create(C.c_uint32(seed))

    def normals(self, n):
        out = np.empty(n, np.float64)
        lib().rd_normals(C.c_void_p(self._h), C.c_int64(n), _p(out))
        return out

    def uniforms(self, n):
        out = np.empty(n, np.float64)
        lib().rd_uniforms(C.c_void_p(self._h), C.c_int64(n), _p(out))
        return out

    def raw(self, n):
        out = np.empty(n, np.uint32)
        lib().rd_raw(C.c_void_p(self._h), C.c_int64(n), _p(out))
        return out

    def __del__(self):
        try:
            lib().rd_destroy(C.c_void_p(self._h))
        except Exception:
            pass


def init_particles_pose(stream: RefStream, pose, N):
    """cpp:382-399."""
    n = stream.normals(3 * N).reshape(N, 3)
    p = np.empty((3, N))
    p[0] = pose[0] + n[:, 0] * 0.5
    p[1] = pose[1] + n[:, 1] * 0.5
    th = pose[2] + n[:, 2] * 0.4
    p[2] = [lib().orc_normalize_angle(float(t)) for t in th]
    return p, np.full(N, 1.0 / N)


# ----------------------------------------------------------------------------- engine spec
def eng_log_table(table, inv_squash=1.0 / 2.2):
    table = _c(table, np.float64)
    P = table.shape[0] - 1
    L = np.empty((P + 1) * (P + 1), np.float32)
    lib().orc_eng_log_table(C.c_int32(P), _p(table), C.c_double(inv_squash), _p(L))
    return L.reshape(P + 1, P + 1)   # L[r_obs, d]


def eng_log_weights(m: OracleMap, p_colmajor, angles, obs_idx, L, want_steps=False, use_omp=True):
    p = _c(p_colmajor, np.float64)
    N = p.shape[1]
    angles = _c(angles, np.float32)
    oi = _c(obs_idx, np.int32)
    L = _c(L, np.float32)
    B = angles.size
    logw = np.empty(N, np.float64)
    steps = np.empty(N * B, np.uint8) if want_steps else None
    probes = C.c_int64(0)
    lib().orc_eng_log_weights(m.ref, C.c_int64(N), _p(p), C.c_int32(B), _p(angles), _p(oi), _p(L), _p(logw),
                              _p(steps), C.byref(probes), int(use_omp))
    return logw, (steps.reshape(N, B) if want_steps else None), int(probes.value)


def eng_det_exp(x):
    return np.array([lib().orc_eng_det_exp(float(v)) for v in np.atleast_1d(x)])


def eng_weights_from_log(logw):
    logw = _c(logw, np.float64)
    w = np.empty(logw.size, np.float64)
    q = np.empty(logw.size, np.uint64)
    mx = lib().orc_eng_weights_from_log(C.c_int64(logw.size), _p(logw), _p(w), _p(q))
    return w, q, mx


def eng_quantize_weights(w):
    w = _c(w, np.float64)
    q = np.empty(w.size, np.uint64)
    lib().orc_eng_quantize_weights(C.c_int64(w.size), _p(w), _p(q))
    return q


def eng_resample_indices(q, mode, n_children=None, k53=None, k0=0):
    q = _c(q, np.uint64)
    n_children = q.size if n_children is None else n_children
    idx = np.empty(n_children, np.int32)
    k = _c(k53, np.uint64) if k53 is not None else None
    lib().orc_eng_resample_indices(C.c_int64(q.size), _p(q), int(mode), C.c_int64(n_children), _p(k),
                                   C.c_uint32(k0), _p(idx))
    return idx


def eng_philox4x32(ctr, key):
    out = np.empty(4, np.uint32)
    lib().orc_eng_philox4x32(*(C.c_uint32(int(c)) for c in ctr), *(C.c_uint32(int(k)) for k in key), _p(out))
    return out


def eng_philox_k53(seed, upd, first, n):
    out = np.empty(n, np.uint64)
    lib().orc_eng_philox_k53(C.c_uint64(seed), C.c_uint32(upd), C.c_int64(first), C.c_int64(n), _p(out))
    return out


def eng_philox_k0(seed, upd):
    return int(lib().orc_eng_philox_k0(C.c_uint64(seed), C.c_uint32(upd)))


def eng_philox_normals(seed, upd, first, n):
    out = np.empty(3 * n, np.float64)
    lib().orc_eng_philox_normals(C.c_uint64(seed), C.c_uint32(upd), C.c_int64(first), C.c_int64(n), _p(out))
    return out.reshape(n, 3)


def eng_init_pose(seed, init_idx, pose, first, n):
    out = np.empty((3, n), np.float64)
    lib().orc_eng_init_pose(C.c_uint64(seed), C.c_uint32(init_idx), _p(_c(pose, np.float64)), C.c_int64(first), C.c_int64(n), _p(out))
    return out


def eng_init_global(seed, init_idx, m: OracleMap, first, n):
    out = np.empty((3, n), np.float64)
    rc = lib().orc_eng_init_global(C.c_uint64(seed), C.c_uint32(init_idx), m.ref, C.c_int64(first), C.c_int64(n), _p(out))
    if rc:
        raise RuntimeError("no free cells")
    return out


def eng_chebyshev_bruteforce(stop, cap):
    stop = _c(stop, np.uint8)
    Hp, Wp = stop.shape
    d = np.empty((Hp, Wp), np.uint8)
    lib().orc_eng_chebyshev_bruteforce(C.c_int32(Wp), C.c_int32(Hp), _p(stop), C.c_int32(cap), _p(d))
    return d
