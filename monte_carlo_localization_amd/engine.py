"""ctypes binding of the C ABI in include/mcl_hip_engine.h (libmcl_hip_engine.so).

This is plumbing only: every numeric step runs in the HIP library.  There is no fallback — if the
shared library is missing or no gfx950 device is visible the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MCL_LIB: another build of the same library, for A/B runs on one box -- a development aid, never set by the product or the tests)
LIB_PATH = os.environ.get("MCL_LIB") or os.path.join(_HERE, "libmcl_hip_engine.so")

MCL_OK = 0
MCL_ERR_NOT_READY = -2
MCL_ERR_PEER = -6          # sharded update: another rank reported a failure; the update is void on every rank
MCL_ERR_TIMEOUT = -7       # sharded update: a collective did not finish in time; the communicator was aborted
RESAMPLE_MULTINOMIAL, RESAMPLE_SYSTEMATIC = 0, 1
WEIGHT_LOG, WEIGHT_PRODUCT = 0, 1
RAYS_AUTO, RAYS_MARCH, RAYS_SKIP, RAYS_QUAD, RAYS_CELL, RAYS_SWEEP = 0, 1, 2, 3, 4, 5
BUF_X, BUF_Y, BUF_THETA, BUF_QWEIGHT, BUF_LOGW, BUF_SCALARS = range(6)

EXPORTS = [
    "mcl_abi_version", "mcl_default_config", "mcl_create", "mcl_destroy", "mcl_last_error", "mcl_set_map",
    "mcl_get_max_range_px", "mcl_get_sensor_table", "mcl_set_beam_angles", "mcl_set_particles",
    "mcl_get_particles", "mcl_get_weights", "mcl_sample_particles", "mcl_particle_mean", "mcl_update",
    "mcl_sensor_update", "mcl_expected_pose", "mcl_get_stage_timings", "mcl_get_resample_indices",
    "mcl_get_ray_steps", "mcl_get_log_weights", "mcl_get_counters", "mcl_get_ray_kernel_ms", "mcl_device_ptr",
    "mcl_stage_propagate", "mcl_stage_weights", "mcl_stage_finish", "mcl_scan_weights", "mcl_export_state",
    "mcl_get_scalars", "mcl_host_sensor_table", "mcl_host_skip_field", "mcl_init_particles_pose", "mcl_init_global",
    "mcl_update_scan", "mcl_get_ray_kernel_id", "mcl_host_skip_field_dir", "mcl_host_skip_field_wedge", "mcl_export_records", "mcl_get_effective_sample_size", "mcl_get_host_scalars", "mcl_stage_resample_records", "mcl_stage_resample", "mcl_stage_rays",
    "mcl_set_reserved_cus", "mcl_stage_resample_indices", "mcl_stage_motion_records", "mcl_stage_distinct_parents", "mcl_export_records_at",
    "mcl_group_create", "mcl_group_destroy", "mcl_group_last_error", "mcl_group_size", "mcl_group_engine", "mcl_group_set_map",
    "mcl_group_set_beam_angles", "mcl_group_set_particles", "mcl_group_init_particles_pose", "mcl_group_init_global",
    "mcl_group_update", "mcl_group_expected_pose", "mcl_group_get_particles", "mcl_group_get_weights",
    "mcl_group_get_resample_indices", "mcl_group_get_stage_timings", "mcl_group_exchange_bytes",
    "mcl_set_debug_count_probes", "mcl_set_particles_shard", "mcl_get_compact_list", "mcl_compact_chunk_bytes", "mcl_export_compact",
    "mcl_stage_resample_compact", "mcl_group_exchanged_lists", "mcl_get_ray_steps16", "mcl_get_planned_ray_kernel",
    "mcl_stream_wait_external", "mcl_external_wait_stream", "mcl_export_compact_async", "mcl_stage_resample_compact_async",
    "mcl_stage_rays_async", "mcl_stage_weights_async", "mcl_stage_complete", "mcl_stage_keep",
    "mcl_comm_available", "mcl_comm_unique_id", "mcl_comm_create", "mcl_comm_destroy", "mcl_comm_update", "mcl_comm_stats", "mcl_comm_set_lists", "mcl_comm_get_vector", "mcl_comm_last_exchange", "mcl_comm_selftest",
    "mcl_host_sweep_global_layout", "mcl_get_ray_kernel_variant",
]


class Config(C.Structure):
    _fields_ = [
        ("max_particles", C.c_int64), ("device", C.c_int32), ("seed", C.c_uint64), ("max_range_m", C.c_double),
        ("z_hit", C.c_double), ("z_short", C.c_double), ("z_max", C.c_double), ("z_rand", C.c_double),
        ("sigma_hit", C.c_double), ("squash_factor", C.c_double), ("motion_dispersion_x", C.c_double),
        ("motion_dispersion_y", C.c_double), ("motion_dispersion_theta", C.c_double), ("resample_mode", C.c_int32),
        ("weight_mode", C.c_int32), ("ray_kernel", C.c_int32), ("keep_ray_steps", C.c_int32),
        ("debug_force_exact", C.c_int32), ("debug_count_probes", C.c_int32), ("rays_per_lane", C.c_int32),
        ("resample_neff_permille", C.c_int32), ("graph_mode", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class EngineError(RuntimeError):
    """A call through the C ABI returned a negative mcl_status; `.status` holds it."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


class ShardedUpdateError(EngineError):
    """An update of a SHARDED set is void on every rank: this rank failed locally (`.local` is True, the message says how) or
    another rank did (MCL_ERR_PEER), or a collective ran out of time and the communicator was aborted (MCL_ERR_TIMEOUT).  Every
    rank raises from the same update; the particle set must be set or initialised again on every rank before the next one."""

    def __init__(self, msg, status=None, local=False):
        super().__init__(msg, status)
        self.local = local


_libs = {}
LEGACY_LIB_PATH = os.path.join(_HERE, "libmcl_hip_engine_legacy.so")


def load_library(legacy=False):
    """Loads libmcl_hip_engine.so (built by __graft_entry__.build()); raises if absent.  legacy=True: the same sources built with
    -DMCL_LEGACY_RAY_KERNELS (libmcl_hip_engine_legacy.so), which also holds k_rays_quad / k_rays_cell -- the predecessors of the
    windowed ray kernel, kept as two more implementations to test it against; an Engine asks for it only when its configuration
    names one of them (ray_kernel = RAYS_QUAD / RAYS_CELL)."""
    path = LEGACY_LIB_PATH if legacy else LIB_PATH
    _lib = _libs.get(path)
    if _lib is None:
        # PyTorch-ROCm wheels bundle their own copies of the ROCm runtime under the same sonames as /opt/rocm's
        # (libamdhip64.so.7, libhsa-runtime64.so.1).  A process that uses both must load torch's first: if this library pulls
        # in /opt/rocm's copies before `import torch`, torch ends up on a runtime it was not built with and reports
        # "No HIP GPUs are available".  So when torch is installed and not imported yet, import it here (dist.py and
        # bench.py need it anyway); without torch the library simply uses /opt/rocm's runtime.
        import importlib.util
        import sys
        if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(path):
            raise EngineError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(path)
        lib.mcl_last_error.restype = C.c_char_p
        lib.mcl_last_error.argtypes = [C.c_void_p]
        lib.mcl_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        lib.mcl_destroy.argtypes = [C.c_void_p]
        lib.mcl_destroy.restype = None
        lib.mcl_default_config.argtypes = [C.POINTER(Config)]
        lib.mcl_default_config.restype = None
        lib.mcl_group_last_error.restype = C.c_char_p
        lib.mcl_group_last_error.argtypes = [C.c_void_p]
        lib.mcl_group_create.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        lib.mcl_group_destroy.argtypes = [C.c_void_p]
        lib.mcl_group_destroy.restype = None
        _libs[path] = _lib = lib
    return _lib


def default_config(**over) -> Config:
    cfg = Config()
    load_library().mcl_default_config(C.byref(cfg))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def host_sensor_table(P: int, cfg: Config | None = None) -> np.ndarray:
    """The engine's host-side sensor table (no device needed); returned as T[d, r]."""
    cfg = cfg or default_config()
    out = np.empty((P + 1) * (P + 1), np.float64)
    rc = load_library().mcl_host_sensor_table(C.byref(cfg), C.c_int32(P), _p(out), C.c_size_t(out.size))
    if rc != MCL_OK:
        raise EngineError(f"mcl_host_sensor_table rc={rc}")
    return out.reshape(P + 1, P + 1)


def host_skip_field(grid) -> np.ndarray:
    """The engine's padded skip-distance field (no device needed): shape (H+1, W+1), uint8."""
    g = _c(grid, np.int8)
    H, W = g.shape
    out = np.empty((H + 1, W + 1), np.uint8)
    rc = load_library().mcl_host_skip_field(_p(g), C.c_uint32(W), C.c_uint32(H), _p(out), C.c_size_t(out.size))
    if rc != MCL_OK:
        raise EngineError(f"mcl_host_skip_field rc={rc}")
    return out


WEDGES = 16


def host_skip_field_wedge(grid, wedge: int) -> np.ndarray:
    """Skip field for rays whose direction angle lies in sector `wedge` of WEDGES equal sectors of the turn."""
    g = _c(grid, np.int8)
    H, W = g.shape
    out = np.empty((H + 1, W + 1), np.uint8)
    rc = load_library().mcl_host_skip_field_wedge(_p(g), C.c_uint32(W), C.c_uint32(H), C.c_int32(wedge), _p(out), C.c_size_t(out.size))
    if rc != MCL_OK:
        raise EngineError(f"mcl_host_skip_field_wedge rc={rc}")
    return out


def host_skip_field_dir(grid, quadrant: int) -> np.ndarray:
    """Directional skip field for rays of one direction quadrant (0:+x+y 1:-x+y 2:-x-y 3:+x-y)."""
    g = _c(grid, np.int8)
    H, W = g.shape
    out = np.empty((H + 1, W + 1), np.uint8)
    rc = load_library().mcl_host_skip_field_dir(_p(g), C.c_uint32(W), C.c_uint32(H), C.c_int32(quadrant), _p(out), C.c_size_t(out.size))
    if rc != MCL_OK:
        raise EngineError(f"mcl_host_skip_field_dir rc={rc}")
    return out


class Engine:
    """One engine == one GPU == one particle shard."""

    def __init__(self, cfg: Config | None = None, **over):
        self.cfg = cfg if cfg is not None else default_config(**over)
        if cfg is not None:
            for k, v in over.items():
                setattr(self.cfg, k, v)
        # (the predecessors of the windowed ray kernel live in the legacy build of the library: load_library)
        self.lib = load_library(legacy=int(self.cfg.ray_kernel) in (RAYS_QUAD, RAYS_CELL))
        h = C.c_void_p()
        rc = self.lib.mcl_create(C.byref(self.cfg), C.byref(h))
        if rc != MCL_OK:
            raise EngineError(f"mcl_create rc={rc}: {self.lib.mcl_last_error(None).decode()}")
        self._h = h
        self.n = 0
        self.n_beams = 0

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.mcl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != MCL_OK:
            raise EngineError(f"{what} rc={rc}: {self.lib.mcl_last_error(self._h).decode()}", rc)

    # -- map / beams
    def set_map(self, grid, resolution, origin_x, origin_y):
        g = _c(grid, np.int8)
        H, W = g.shape
        self._chk(self.lib.mcl_set_map(self._h, _p(g), C.c_uint32(W), C.c_uint32(H), C.c_float(np.float32(resolution)),
                                       C.c_double(origin_x), C.c_double(origin_y)), "mcl_set_map")

    @property
    def max_range_px(self) -> int:
        v = C.c_int32()
        self._chk(self.lib.mcl_get_max_range_px(self._h, C.byref(v)), "mcl_get_max_range_px")
        return v.value

    def sensor_table(self):
        P = self.max_range_px
        out = np.empty((P + 1) * (P + 1), np.float64)
        self._chk(self.lib.mcl_get_sensor_table(self._h, _p(out), C.c_size_t(out.size)), "mcl_get_sensor_table")
        return out.reshape(P + 1, P + 1)      # [d, r]

    def set_beam_angles(self, angles):
        a = _c(angles, np.float32)
        self._chk(self.lib.mcl_set_beam_angles(self._h, _p(a), C.c_int32(a.size)), "mcl_set_beam_angles")
        self.n_beams = a.size

    # -- particles
    def set_particles(self, xyz_colmajor, weights):
        p = _c(xyz_colmajor, np.float64)
        assert p.ndim == 2 and p.shape[0] == 3
        w = _c(weights, np.float64)
        n = p.shape[1]
        assert w.size == n
        self._chk(self.lib.mcl_set_particles(self._h, _p(p), _p(w), C.c_int64(n)), "mcl_set_particles")
        self.n = n

    def set_particles_shard(self, xyz_colmajor, weights, max_weight_of_the_whole_set):
        """set_particles for one shard of a larger set: weights are quantised against the whole set's maximum weight."""
        p = _c(xyz_colmajor, np.float64)
        w = _c(weights, np.float64)
        n = p.shape[1]
        assert p.ndim == 2 and p.shape[0] == 3 and w.size == n
        self._chk(self.lib.mcl_set_particles_shard(self._h, _p(p), _p(w), C.c_int64(n), C.c_double(max_weight_of_the_whole_set)),
                  "mcl_set_particles_shard")
        self.n = n

    def init_particles_pose(self, pose, n, first_global_index=0, n_total=None):
        p = _c(pose, np.float64)
        self._chk(self.lib.mcl_init_particles_pose(self._h, _p(p), C.c_int64(n), C.c_int64(first_global_index),
                                                   C.c_int64(n_total or n)), "mcl_init_particles_pose")
        self.n = n

    def init_global(self, n, first_global_index=0, n_total=None):
        self._chk(self.lib.mcl_init_global(self._h, C.c_int64(n), C.c_int64(first_global_index), C.c_int64(n_total or n)),
                  "mcl_init_global")
        self.n = n

    def update_scan(self, action, ranges, angle_step):
        a, r = _c(action, np.float64), _c(ranges, np.float32)
        self._chk(self.lib.mcl_update_scan(self._h, _p(a), _p(r), C.c_int32(r.size), C.c_int32(angle_step)), "mcl_update_scan")

    def get_particles(self):
        out = np.empty((3, self.n), np.float64)
        self._chk(self.lib.mcl_get_particles(self._h, _p(out), C.c_int64(self.n)), "mcl_get_particles")
        return out

    def get_weights(self):
        out = np.empty(self.n, np.float64)
        self._chk(self.lib.mcl_get_weights(self._h, _p(out), C.c_int64(self.n)), "mcl_get_weights")
        return out

    def sample_particles(self, k, uniforms=None):
        u = _c(uniforms, np.float64)
        out = np.empty((3, k), np.float64)
        self._chk(self.lib.mcl_sample_particles(self._h, C.c_int32(k), _p(u), _p(out)), "mcl_sample_particles")
        return out

    def particle_mean(self):
        out = np.empty(3)
        self._chk(self.lib.mcl_particle_mean(self._h, _p(out)), "mcl_particle_mean")
        return out

    # -- update
    def update(self, action, obs, normals=None, uniforms=None):
        a = _c(action, np.float64)
        o = _c(obs, np.float32)
        nrm, u = _c(normals, np.float64), _c(uniforms, np.float64)
        if nrm is not None:
            assert nrm.size == 3 * self.n
        if u is not None:
            assert u.size == self.n
        self._chk(self.lib.mcl_update(self._h, _p(a), _p(o), C.c_int32(o.size), _p(nrm), _p(u)), "mcl_update")

    def sensor_update(self, obs):
        o = _c(obs, np.float32)
        self._chk(self.lib.mcl_sensor_update(self._h, _p(o), C.c_int32(o.size)), "mcl_sensor_update")

    def expected_pose(self):
        out = np.empty(3)
        self._chk(self.lib.mcl_expected_pose(self._h, _p(out)), "mcl_expected_pose")
        return out

    def stage_timings(self):
        out = np.empty(6)
        self._chk(self.lib.mcl_get_stage_timings(self._h, _p(out)), "mcl_get_stage_timings")
        return out

    # -- diagnostics
    def resample_indices(self):
        out = np.empty(self.n, np.int32)
        self._chk(self.lib.mcl_get_resample_indices(self._h, _p(out), C.c_int64(self.n)), "mcl_get_resample_indices")
        return out

    def ray_steps(self):
        """Step index of every ray of the last update (needs keep_ray_steps): uint8 up to 255 px of range, uint16 beyond."""
        if self.max_range_px > 255:
            out = np.empty(self.n * self.n_beams, np.uint16)
            self._chk(self.lib.mcl_get_ray_steps16(self._h, _p(out), C.c_size_t(out.size)), "mcl_get_ray_steps16")
        else:
            out = np.empty(self.n * self.n_beams, np.uint8)
            self._chk(self.lib.mcl_get_ray_steps(self._h, _p(out), C.c_size_t(out.size)), "mcl_get_ray_steps")
        return out.reshape(self.n, self.n_beams)

    def log_weights(self):
        out = np.empty(self.n, np.float64)
        self._chk(self.lib.mcl_get_log_weights(self._h, _p(out), C.c_int64(self.n)), "mcl_get_log_weights")
        return out

    def counters(self):
        out = np.zeros(4, np.uint64)
        self._chk(self.lib.mcl_get_counters(self._h, _p(out)), "mcl_get_counters")
        return dict(exact_fallback_rays=int(out[0]), off_window_particles=int(out[1]), probes=int(out[2]),
                    level2_rays=int(out[3]))

    def compact_list(self):
        """(entries of the compact parent list that describes the current weights or -1, whether the last resampling used one)"""
        n, u = C.c_int64(), C.c_int32()
        self._chk(self.lib.mcl_get_compact_list(self._h, C.byref(n), C.byref(u)), "mcl_get_compact_list")
        return n.value, bool(u.value)

    def set_debug_count_probes(self, on):
        self._chk(self.lib.mcl_set_debug_count_probes(self._h, C.c_int32(1 if on else 0)), "mcl_set_debug_count_probes")

    def ray_kernel_ms(self):
        v = C.c_double()
        self._chk(self.lib.mcl_get_ray_kernel_ms(self._h, C.byref(v)), "mcl_get_ray_kernel_ms")
        return v.value

    def effective_sample_size(self):
        """(N_eff of the current weights, whether the last update resampled)"""
        v, r = C.c_double(), C.c_int32()
        self._chk(self.lib.mcl_get_effective_sample_size(self._h, C.byref(v), C.byref(r)), "mcl_get_effective_sample_size")
        return v.value, bool(r.value)

    def ray_kernel_name(self):
        v = C.c_int32()
        self._chk(self.lib.mcl_get_ray_kernel_id(self._h, C.byref(v)), "mcl_get_ray_kernel_id")
        return {1: "k_rays_march", 2: "k_rays_skip", 3: "k_rays_quad", 4: "k_rays_cell", 5: "k_rays_sweep"}.get(v.value, "?")

    def ray_kernel_variant(self):
        """Form of k_rays_sweep the last ray stage ran: dict(global_fields, hybrid, turned_directions, pairs)."""
        v = (C.c_int32 * 3)()
        self._chk(self.lib.mcl_get_ray_kernel_variant(self._h, v), "mcl_get_ray_kernel_variant")
        return dict(global_fields=v[0] == 1, hybrid=v[0] == 2, turned_directions=bool(v[1]), pairs=bool(v[2]))

    RAY_KERNEL_NAMES = {0: None, 1: "k_rays_march", 2: "k_rays_skip", 3: "k_rays_quad", 4: "k_rays_cell", 5: "k_rays_sweep"}

    def planned_ray_kernel(self, n_particles=0):
        """(kernel an update over n_particles WILL run -- None: the configured one cannot run with this map / beam set --, what
        decided): callable before the first update, once the map and the beam angles are set."""
        v, why = C.c_int32(), C.c_char_p()
        self._chk(self.lib.mcl_get_planned_ray_kernel(self._h, C.c_int64(int(n_particles)), C.byref(v), C.byref(why)), "mcl_get_planned_ray_kernel")
        return self.RAY_KERNEL_NAMES.get(v.value, "?"), (why.value or b"").decode()

    # -- multi-GPU staging (raw device pointers as ints)
    def device_ptr(self, which) -> int:
        p = C.c_void_p()
        self._chk(self.lib.mcl_device_ptr(self._h, C.c_int32(which), C.byref(p)), "mcl_device_ptr")
        return int(p.value or 0)

    def export_state(self, d_x=0, d_y=0, d_th=0, d_q=0):
        self._chk(self.lib.mcl_export_state(self._h, C.c_void_p(d_x or None), C.c_void_p(d_y or None),
                                            C.c_void_p(d_th or None), C.c_void_p(d_q or None)), "mcl_export_state")

    def scalars(self):
        out = np.empty(8)
        self._chk(self.lib.mcl_get_scalars(self._h, _p(out)), "mcl_get_scalars")
        return out

    def host_scalars(self):
        """SCALARS as the last stage call read them back (no device access)."""
        out = np.empty(8)
        self._chk(self.lib.mcl_get_host_scalars(self._h, _p(out)), "mcl_get_host_scalars")
        return out

    def stage_propagate(self, d_px, d_py, d_pth, d_cdf, n_parents, q_total, child_first, n_children_total, action, obs):
        a = _c(action, np.float64)
        o = _c(obs, np.float32)
        self._chk(self.lib.mcl_stage_propagate(self._h, C.c_void_p(d_px), C.c_void_p(d_py), C.c_void_p(d_pth),
                                               C.c_void_p(d_cdf), C.c_int64(n_parents), C.c_uint64(q_total),
                                               C.c_int64(child_first), C.c_int64(n_children_total), _p(a), _p(o),
                                               C.c_int32(o.size)), "mcl_stage_propagate")

    def stage_resample(self, d_px, d_py, d_pth, d_cdf, n_parents, q_total, child_first, n_children_total, action):
        a = _c(action, np.float64)
        self._chk(self.lib.mcl_stage_resample(self._h, C.c_void_p(d_px), C.c_void_p(d_py), C.c_void_p(d_pth),
                                              C.c_void_p(d_cdf), C.c_int64(n_parents), C.c_uint64(q_total),
                                              C.c_int64(child_first), C.c_int64(n_children_total), _p(a)), "mcl_stage_resample")

    def export_records(self, d_records):
        self._chk(self.lib.mcl_export_records(self._h, C.c_void_p(d_records)), "mcl_export_records")

    def stage_resample_records(self, d_records, d_cdf, n_parents, q_total, child_first, n_children_total, action):
        a = _c(action, np.float64)
        self._chk(self.lib.mcl_stage_resample_records(self._h, C.c_void_p(d_records), C.c_void_p(d_cdf), C.c_int64(n_parents),
                                                      C.c_uint64(q_total), C.c_int64(child_first), C.c_int64(n_children_total), _p(a)),
                  "mcl_stage_resample_records")

    def stage_resample_indices(self, d_cdf, n_parents, q_total, child_first, n_children_total, d_parent_idx):
        self._chk(self.lib.mcl_stage_resample_indices(self._h, C.c_void_p(d_cdf), C.c_int64(n_parents), C.c_uint64(q_total),
                                                      C.c_int64(child_first), C.c_int64(n_children_total), C.c_void_p(d_parent_idx)),
                  "mcl_stage_resample_indices")

    def stage_distinct_parents(self, d_parent, n_children, n_total, d_distinct, d_slot) -> int:
        """Distinct parents (ascending) of `n_children` global parent indices + every child's position among them; returns
        their number.  All pointers are device memory (int32 in, int64 / int32 out)."""
        cnt = C.c_int64(0)
        self._chk(self.lib.mcl_stage_distinct_parents(self._h, C.c_void_p(d_parent), C.c_int64(n_children), C.c_int64(n_total),
                                                      C.c_void_p(d_distinct), C.c_void_p(d_slot), C.byref(cnt)), "mcl_stage_distinct_parents")
        return int(cnt.value)

    def export_records_at(self, d_index, count, d_out):
        """Packed records of the listed local particles (device int64 indices) -> d_out[count] (device)."""
        self._chk(self.lib.mcl_export_records_at(self._h, C.c_void_p(d_index), C.c_int64(count), C.c_void_p(d_out)), "mcl_export_records_at")

    def stage_motion_records(self, d_records, n_records, d_record_of_child, child_first, n_children_total, action):
        a = _c(action, np.float64)
        self._chk(self.lib.mcl_stage_motion_records(self._h, C.c_void_p(d_records), C.c_int64(n_records), C.c_void_p(d_record_of_child),
                                                    C.c_int64(child_first), C.c_int64(n_children_total), _p(a)), "mcl_stage_motion_records")

    @staticmethod
    def compact_chunk_bytes(chunk_entries) -> int:
        b = C.c_int64()
        rc = load_library().mcl_compact_chunk_bytes(C.c_int64(chunk_entries), C.byref(b))
        if rc != MCL_OK:
            raise EngineError(f"mcl_compact_chunk_bytes rc={rc}")
        return b.value

    def export_compact(self, d_chunk, chunk_entries):
        """This shard's compact parent list -> d_chunk (device memory of compact_chunk_bytes(chunk_entries))."""
        self._chk(self.lib.mcl_export_compact(self._h, C.c_void_p(d_chunk), C.c_int64(chunk_entries)), "mcl_export_compact")

    def stage_resample_compact(self, d_chunks, n_shards, chunk_entries, counts, totals, n_per_shard, self_shard, child_first, n_children_total, action):
        a = _c(action, np.float64)
        c = np.ascontiguousarray(np.asarray(counts, np.int64))
        t = np.ascontiguousarray(np.asarray(totals, np.uint64))
        assert c.size == n_shards and t.size == n_shards
        self._chk(self.lib.mcl_stage_resample_compact(self._h, C.c_void_p(d_chunks), C.c_int32(n_shards), C.c_int64(chunk_entries), _p(c), _p(t),
                                                      C.c_int64(n_per_shard), C.c_int32(self_shard), C.c_int64(child_first),
                                                      C.c_int64(n_children_total), _p(a)), "mcl_stage_resample_compact")

    def stage_rays(self, obs):
        o = _c(obs, np.float32)
        self._chk(self.lib.mcl_stage_rays(self._h, _p(o), C.c_int32(o.size)), "mcl_stage_rays")

    def set_reserved_cus(self, n):
        self._chk(self.lib.mcl_set_reserved_cus(self._h, C.c_int32(n)), "mcl_set_reserved_cus")

    def stage_weights(self, global_max):
        self._chk(self.lib.mcl_stage_weights(self._h, C.c_double(global_max)), "mcl_stage_weights")

    def stage_finish(self, sums5):
        s = _c(sums5, np.float64)
        self._chk(self.lib.mcl_stage_finish(self._h, _p(s)), "mcl_stage_finish")

    # ---- the staged flow ordered on the device (include/mcl_hip_engine.h: "ORDERED ON THE DEVICE"): nothing here waits for the
    # stream except stage_complete; `stream` is a raw hipStream_t (torch.cuda.current_stream().cuda_stream)
    def stream_wait_external(self, stream):
        self._chk(self.lib.mcl_stream_wait_external(self._h, C.c_void_p(stream)), "mcl_stream_wait_external")

    def external_wait_stream(self, stream):
        self._chk(self.lib.mcl_external_wait_stream(self._h, C.c_void_p(stream)), "mcl_external_wait_stream")

    def export_compact_async(self, d_chunk, chunk_entries):
        self._chk(self.lib.mcl_export_compact_async(self._h, C.c_void_p(d_chunk), C.c_int64(chunk_entries)), "mcl_export_compact_async")

    def stage_resample_compact_async(self, d_chunks, n_shards, chunk_entries, counts, totals, n_per_shard, self_shard, child_first, n_children_total,
                                     action):
        a = _c(action, np.float64)
        c = np.ascontiguousarray(np.asarray(counts, np.int64))
        t = np.ascontiguousarray(np.asarray(totals, np.uint64))
        assert c.size == n_shards and t.size == n_shards
        self._chk(self.lib.mcl_stage_resample_compact_async(self._h, C.c_void_p(d_chunks), C.c_int32(n_shards), C.c_int64(chunk_entries), _p(c), _p(t),
                                                            C.c_int64(n_per_shard), C.c_int32(self_shard), C.c_int64(child_first),
                                                            C.c_int64(n_children_total), _p(a)), "mcl_stage_resample_compact_async")

    def stage_keep(self, child_first, n_children_total, action):
        """Adaptive resampling kept the set: motion only, every particle its own parent (launch only)."""
        a = _c(action, np.float64)
        self._chk(self.lib.mcl_stage_keep(self._h, C.c_int64(child_first), C.c_int64(n_children_total), _p(a)), "mcl_stage_keep")

    def stage_rays_async(self, obs, d_local_max):
        o = _c(obs, np.float32)
        self._chk(self.lib.mcl_stage_rays_async(self._h, _p(o), C.c_int32(o.size), C.c_void_p(d_local_max)), "mcl_stage_rays_async")

    def stage_weights_async(self, d_global_max, d_vec, n_shards, self_shard):
        self._chk(self.lib.mcl_stage_weights_async(self._h, C.c_void_p(d_global_max), C.c_void_p(d_vec), C.c_int32(n_shards), C.c_int32(self_shard)),
                  "mcl_stage_weights_async")

    def stage_complete(self, sums5) -> bool:
        """The one host wait of a device-ordered update.  True: the ray stage's fix-up lists overflowed, run the synchronous
        stages once more (stage_rays .. stage_finish)."""
        s = _c(sums5, np.float64)
        redo = C.c_int32(0)
        self._chk(self.lib.mcl_stage_complete(self._h, _p(s), C.byref(redo)), "mcl_stage_complete")
        return bool(redo.value)

    # ---- the sharded update in native code: an RCCL communicator inside the engine (include/mcl_hip_engine.h: mcl_comm_*)
    @staticmethod
    def comm_available():
        """(True, "") when the library finds an RCCL to use, else (False, why)."""
        why = C.c_char_p()
        rc = load_library().mcl_comm_available(C.byref(why))
        return rc == MCL_OK, (why.value or b"").decode()

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_ubyte * 128)()
        rc = load_library().mcl_comm_unique_id(buf)
        if rc != MCL_OK:
            raise EngineError(f"mcl_comm_unique_id rc={rc}")
        return bytes(buf)

    def comm_create(self, uid: bytes, n_ranks: int, rank: int):
        """COLLECTIVE: every rank calls it with rank 0's id."""
        assert len(uid) == 128
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        self._chk(self.lib.mcl_comm_create(self._h, buf, C.c_int32(n_ranks), C.c_int32(rank)), "mcl_comm_create")
        self._comm_ranks = n_ranks

    def comm_selftest(self):
        """COLLECTIVE: the three collectives of an update on known data."""
        self._chk(self.lib.mcl_comm_selftest(self._h), "mcl_comm_selftest")

    def comm_destroy(self):
        self._chk(self.lib.mcl_comm_destroy(self._h), "mcl_comm_destroy")

    def comm_set_lists(self, counts, totals):
        c = np.ascontiguousarray(np.asarray(counts, np.int64))
        t = np.ascontiguousarray(np.asarray(totals, np.uint64))
        assert c.size == self._comm_ranks and t.size == self._comm_ranks
        self._chk(self.lib.mcl_comm_set_lists(self._h, _p(c), _p(t)), "mcl_comm_set_lists")

    def comm_update(self, action, obs):
        """One sharded update in one native call (lists, or the dense exchange when there are none); the pose of the whole set.
        Any failure -- this rank's, a peer's, a collective that ran out of time -- raises ShardedUpdateError on EVERY rank from
        the same update (include/mcl_hip_engine.h: the failure protocol of mcl_comm_update); nothing falls back to another exchange."""
        a = _c(action, np.float64)
        o = _c(obs, np.float32)
        pose = np.zeros(3)
        rc = self.lib.mcl_comm_update(self._h, _p(a), _p(o), C.c_int32(o.size), _p(pose))
        if rc != MCL_OK:
            raise ShardedUpdateError(f"mcl_comm_update rc={rc}: {self.lib.mcl_last_error(self._h).decode()}", rc,
                                     local=rc not in (MCL_ERR_PEER, MCL_ERR_TIMEOUT))
        return pose

    def comm_vector(self):
        vec = np.zeros(5 + 3 * self._comm_ranks + 2)
        self._chk(self.lib.mcl_comm_get_vector(self._h, _p(vec), C.c_int32(vec.size)), "mcl_comm_get_vector")
        return vec

    def comm_stats(self):
        r, p, w = C.c_uint64(), C.c_uint64(), C.c_int32()
        self._chk(self.lib.mcl_comm_stats(self._h, C.byref(r), C.byref(p), C.byref(w)), "mcl_comm_stats")
        d, wb, rb = C.c_int32(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.mcl_comm_last_exchange(self._h, C.byref(d), C.byref(wb), C.byref(rb)), "mcl_comm_last_exchange")
        return dict(list_bytes_received=r.value, list_payload_bytes=p.value, host_waits=w.value, dense=d.value == 1, kept=d.value == 2,
                    weights_received=wb.value, records_received=rb.value)

    def scan_weights(self, d_q, d_cdf, n, offset=0):
        self._chk(self.lib.mcl_scan_weights(self._h, C.c_void_p(d_q), C.c_void_p(d_cdf), C.c_int64(n),
                                            C.c_uint64(offset)), "mcl_scan_weights")


class Group:
    """Several GPUs behind one handle, driven by this one process (mcl_group_*): the particle set is sharded contiguously
    over `devices`; results are bit-identical to one Engine holding all particles."""

    def __init__(self, devices, cfg: Config | None = None, **over):
        self.lib = load_library()
        self.cfg = cfg if cfg is not None else default_config(**over)
        if cfg is not None:
            for k, v in over.items():
                setattr(self.cfg, k, v)
        dev = np.ascontiguousarray(np.asarray(devices, np.int32))
        h = C.c_void_p()
        rc = self.lib.mcl_group_create(C.byref(self.cfg), _p(dev), C.c_int32(dev.size), C.byref(h))
        if rc != MCL_OK:
            raise EngineError(f"mcl_group_create rc={rc}: {self.lib.mcl_group_last_error(None).decode()}")
        self._h = h
        self.size = int(dev.size)
        self.n_total = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mcl_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != MCL_OK:
            raise EngineError(f"{what} rc={rc}: {self.lib.mcl_group_last_error(self._h).decode()}")

    def set_map(self, grid, resolution, origin_x, origin_y):
        g = _c(grid, np.int8)
        H, W = g.shape
        self._chk(self.lib.mcl_group_set_map(self._h, _p(g), C.c_uint32(W), C.c_uint32(H), C.c_float(resolution),
                                             C.c_double(origin_x), C.c_double(origin_y)), "mcl_group_set_map")

    def set_beam_angles(self, angles):
        a = _c(angles, np.float32)
        self._chk(self.lib.mcl_group_set_beam_angles(self._h, _p(a), C.c_int32(a.size)), "mcl_group_set_beam_angles")

    def set_particles(self, p_colmajor, weights):
        p = _c(p_colmajor, np.float64)
        w = _c(weights, np.float64)
        self.n_total = int(p.shape[1])
        self._chk(self.lib.mcl_group_set_particles(self._h, _p(p), _p(w), C.c_int64(self.n_total)), "mcl_group_set_particles")

    def init_particles_pose(self, pose, n_total):
        q = _c(pose, np.float64)
        self.n_total = int(n_total)
        self._chk(self.lib.mcl_group_init_particles_pose(self._h, _p(q), C.c_int64(n_total)), "mcl_group_init_particles_pose")

    def init_global(self, n_total):
        self.n_total = int(n_total)
        self._chk(self.lib.mcl_group_init_global(self._h, C.c_int64(n_total)), "mcl_group_init_global")

    def update(self, action, obs):
        a = _c(action, np.float64)
        o = _c(obs, np.float32)
        self._chk(self.lib.mcl_group_update(self._h, _p(a), _p(o), C.c_int32(o.size)), "mcl_group_update")

    def expected_pose(self):
        out = np.empty(3)
        self._chk(self.lib.mcl_group_expected_pose(self._h, _p(out)), "mcl_group_expected_pose")
        return out

    def get_particles(self):
        out = np.empty((3, self.n_total))
        self._chk(self.lib.mcl_group_get_particles(self._h, _p(out), C.c_int64(self.n_total)), "mcl_group_get_particles")
        return out

    def get_weights(self):
        out = np.empty(self.n_total)
        self._chk(self.lib.mcl_group_get_weights(self._h, _p(out), C.c_int64(self.n_total)), "mcl_group_get_weights")
        return out

    def resample_indices(self):
        out = np.empty(self.n_total, np.int32)
        self._chk(self.lib.mcl_group_get_resample_indices(self._h, _p(out), C.c_int64(self.n_total)), "mcl_group_get_resample_indices")
        return out

    def stage_timings(self):
        out = np.empty(6)
        self._chk(self.lib.mcl_group_get_stage_timings(self._h, _p(out)), "mcl_group_get_stage_timings")
        return out

    def engine(self, i) -> "Engine":
        """Non-owning view of shard i's engine (diagnostics: sample_particles, counters, ...)."""
        h = C.c_void_p()
        self._chk(self.lib.mcl_group_engine(self._h, C.c_int32(i), C.byref(h)), "mcl_group_engine")
        e = Engine.__new__(Engine)
        e.lib, e.cfg, e._h, e.n, e.n_beams, e._borrowed = self.lib, self.cfg, h, self.n_total // self.size, 0, True
        return e

    def exchange_bytes(self):
        out = np.zeros(2, np.uint64)
        self._chk(self.lib.mcl_group_exchange_bytes(self._h, _p(out)), "mcl_group_exchange_bytes")
        return dict(weights_received_per_device=int(out[0]), parent_records_from_peers=int(out[1]),
                    lists=bool(self.lib.mcl_group_exchanged_lists(self._h)))
