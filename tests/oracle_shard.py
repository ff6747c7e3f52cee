"""CPU stand-in for engine.Engine's multi-GPU staging interface, built on the oracle's engine-spec
functions.  Lives under tests/ on purpose: it lets the gloo world_size-2 tests drive the REAL
orchestration code (monte_carlo_localization_amd/dist.py) on a box without a GPU.  Never shipped."""
import ctypes

import numpy as np

from oracle import oracle as orc


def _view(ptr, n, ctype, dtype):
    if not ptr:
        return None
    return np.frombuffer((ctype * n).from_address(ptr), dtype=dtype)


class OracleShard:
    def __init__(self, om, angles, seed, resample_mode=0):
        self.om, self.angles, self.seed, self.mode = om, np.asarray(angles, np.float32), int(seed), resample_mode
        self.T = orc.sensor_table(om.max_range_px)
        self.L = orc.eng_log_table(self.T)
        self.upd = 0
        self._scalars = np.zeros(8)
        self.idx = None

    def set_particles(self, p, w):
        self.p = np.array(p, np.float64)
        self.n = self.p.shape[1]
        self.w = np.array(w, np.float64)
        self.q = orc.eng_quantize_weights(self.w)

    # ---- staging interface (same names/arguments as engine.Engine) ----
    def export_state(self, d_x=0, d_y=0, d_th=0, d_q=0):
        n = self.n
        for ptr, src in ((d_x, self.p[0]), (d_y, self.p[1]), (d_th, self.p[2])):
            if ptr:
                _view(ptr, n, ctypes.c_double, np.float64)[:] = src
        if d_q:
            _view(d_q, n, ctypes.c_uint64, np.uint64)[:] = self.q

    def export_records(self, d_records):
        v = _view(d_records, 4 * self.n, ctypes.c_double, np.float64).reshape(self.n, 4)
        v[:, 0], v[:, 1], v[:, 2], v[:, 3] = self.p[0], self.p[1], self.p[2], 0.0

    def stage_resample_records(self, d_records, d_cdf, n_parents, q_total, child_first, n_children_total, action):
        rec = _view(d_records, 4 * n_parents, ctypes.c_double, np.float64).reshape(n_parents, 4)
        cols = [np.ascontiguousarray(rec[:, k]) for k in range(3)]
        self.stage_resample(cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, d_cdf, n_parents, q_total, child_first,
                            n_children_total, action)

    def scan_weights(self, d_q, d_cdf, n, offset=0):
        q = _view(d_q, n, ctypes.c_uint64, np.uint64)
        _view(d_cdf, n, ctypes.c_uint64, np.uint64)[:] = np.cumsum(q, dtype=np.uint64) + np.uint64(offset)

    def stage_propagate(self, d_px, d_py, d_pth, d_cdf, n_parents, q_total, child_first, n_children_total, action, obs):
        self.stage_resample(d_px, d_py, d_pth, d_cdf, n_parents, q_total, child_first, n_children_total, action)
        self.stage_rays(obs)

    def stage_resample(self, d_px, d_py, d_pth, d_cdf, n_parents, q_total, child_first, n_children_total, action):
        px = _view(d_px, n_parents, ctypes.c_double, np.float64)
        py = _view(d_py, n_parents, ctypes.c_double, np.float64)
        pth = _view(d_pth, n_parents, ctypes.c_double, np.float64)
        cdf = _view(d_cdf, n_parents, ctypes.c_uint64, np.uint64)
        assert int(cdf[-1]) == q_total
        q = np.diff(cdf, prepend=np.uint64(0)).astype(np.uint64)
        n = self.n
        if self.mode == 0:
            k53_all = orc.eng_philox_k53(self.seed, self.upd, child_first, n)
            idx = orc.eng_resample_indices(q, 0, n_children=n, k53=k53_all)
        else:
            full = orc.eng_resample_indices(q, 1, n_children=n_children_total, k0=orc.eng_philox_k0(self.seed, self.upd))
            idx = full[child_first:child_first + n]
        self.idx = idx
        parents = np.stack([px[idx], py[idx], pth[idx]])
        nrm = orc.eng_philox_normals(self.seed, self.upd, child_first, n)
        self.p = orc.motion_model(parents, action, nrm)
        self.upd += 1

    def _draw(self, d_cdf, n_parents, q_total, child_first, n_children_total):
        cdf = _view(d_cdf, n_parents, ctypes.c_uint64, np.uint64)
        assert int(cdf[-1]) == q_total
        q = np.diff(cdf, prepend=np.uint64(0)).astype(np.uint64)
        n = self.n
        if self.mode == 0:
            return orc.eng_resample_indices(q, 0, n_children=n, k53=orc.eng_philox_k53(self.seed, self.upd, child_first, n))
        full = orc.eng_resample_indices(q, 1, n_children=n_children_total, k0=orc.eng_philox_k0(self.seed, self.upd))
        return full[child_first:child_first + n]

    def stage_resample_indices(self, d_cdf, n_parents, q_total, child_first, n_children_total, d_parent_idx):
        """global parent of every local child -> caller's int32 buffer; the shard's particles stay as they are"""
        self.idx = self._draw(d_cdf, n_parents, q_total, child_first, n_children_total)
        _view(d_parent_idx, self.n, ctypes.c_int32, np.int32)[:] = self.idx

    def stage_distinct_parents(self, d_parent, n_children, n_total, d_distinct, d_slot):
        par = _view(d_parent, n_children, ctypes.c_int32, np.int32)
        uniq, inv = np.unique(par, return_inverse=True)
        _view(d_distinct, n_children, ctypes.c_int64, np.int64)[:uniq.size] = uniq
        _view(d_slot, n_children, ctypes.c_int32, np.int32)[:] = inv.astype(np.int32)
        return int(uniq.size)

    def export_records_at(self, d_index, count, d_out):
        idx = _view(d_index, count, ctypes.c_int64, np.int64)
        out = _view(d_out, 4 * count, ctypes.c_double, np.float64).reshape(count, 4)
        out[:, 0] = self.p[0, idx]; out[:, 1] = self.p[1, idx]; out[:, 2] = self.p[2, idx]; out[:, 3] = 0.0

    def stage_motion_records(self, d_records, n_records, d_record_of_child, child_first, n_children_total, action):
        rec = _view(d_records, 4 * n_records, ctypes.c_double, np.float64).reshape(n_records, 4)
        slot = _view(d_record_of_child, self.n, ctypes.c_int32, np.int32)
        parents = np.stack([rec[slot, 0], rec[slot, 1], rec[slot, 2]])
        nrm = orc.eng_philox_normals(self.seed, self.upd, child_first, self.n)
        self.p = orc.motion_model(parents, action, nrm)
        self.upd += 1

    # ---- compact parent lists (mcl_get_compact_list / mcl_export_compact / mcl_stage_resample_compact)
    def compact_list(self):
        alive = np.nonzero(self.q)[0]
        cap = max(4096, (self.n // 4 + 63) // 64 * 64)
        return (int(alive.size) if alive.size <= cap else -1), False

    def export_compact(self, d_chunk, chunk_entries):
        alive = np.nonzero(self.q)[0]
        k = alive.size
        assert k <= chunk_entries and chunk_entries % 64 == 0
        cdf = np.cumsum(self.q, dtype=np.uint64)
        _view(d_chunk, chunk_entries, ctypes.c_uint64, np.uint64)[:k] = cdf[alive]
        rec = _view(d_chunk + 8 * chunk_entries, 4 * chunk_entries, ctypes.c_double, np.float64).reshape(chunk_entries, 4)
        rec[:k, 0], rec[:k, 1], rec[:k, 2], rec[:k, 3] = self.p[0, alive], self.p[1, alive], self.p[2, alive], 0.0
        _view(d_chunk + 40 * chunk_entries, chunk_entries, ctypes.c_uint32, np.uint32)[:k] = alive.astype(np.uint32)

    def stage_resample_compact(self, d_chunks, n_shards, chunk_entries, counts, totals, n_per_shard, self_shard, child_first, n_children_total, action):
        """The scalar statement of the list exchange: the lists are the whole weighted population -- rebuild the sparse global
        weight vector from them and draw exactly as from the dense one."""
        q = np.zeros(n_shards * n_per_shard, np.uint64)
        rec = np.zeros((n_shards * n_per_shard, 3))
        for r in range(n_shards):
            k = int(counts[r])
            if k == 0:
                continue
            base = d_chunks + 44 * chunk_entries * r
            cdf = _view(base, chunk_entries, ctypes.c_uint64, np.uint64)[:k]
            assert int(cdf[-1]) == int(totals[r])
            idx = _view(base + 40 * chunk_entries, chunk_entries, ctypes.c_uint32, np.uint32)[:k].astype(np.int64) + r * n_per_shard
            q[idx] = np.diff(cdf, prepend=np.uint64(0)).astype(np.uint64)
            rec[idx] = _view(base + 8 * chunk_entries, 4 * chunk_entries, ctypes.c_double, np.float64).reshape(chunk_entries, 4)[:k, :3]
        n = self.n
        if self.mode == 0:
            idx = orc.eng_resample_indices(q, 0, n_children=n, k53=orc.eng_philox_k53(self.seed, self.upd, child_first, n))
        else:
            idx = orc.eng_resample_indices(q, 1, n_children=n_children_total, k0=orc.eng_philox_k0(self.seed, self.upd))[child_first:child_first + n]
        self.idx = idx
        assert (q[idx] > 0).all()
        nrm = orc.eng_philox_normals(self.seed, self.upd, child_first, n)
        self.p = orc.motion_model(np.ascontiguousarray(rec[idx].T), action, nrm)
        self.upd += 1

    def set_particles_shard(self, p, w, max_weight):
        self.p = np.array(p, np.float64)
        self.n = self.p.shape[1]
        self.w = np.array(w, np.float64)
        self.q = np.floor(np.where(self.w > 0, self.w / max_weight, 0.0) * 2.0 ** 36).astype(np.uint64)

    def stage_rays(self, obs):
        oi = orc.obs_index(np.asarray(obs, np.float32), self.om)
        self.logw, _, _ = orc.eng_log_weights(self.om, self.p, self.angles, oi, self.L, use_omp=False)
        self._scalars[0] = self.logw.max()

    def scalars(self):
        return self._scalars.copy()

    def stage_weights(self, global_max):
        w = orc.eng_det_exp(self.logw - global_max)
        self.w = w
        self.q = np.floor(w * 2.0 ** 36).astype(np.uint64)
        s = self._scalars
        s[0] = global_max
        s[1] = w.sum()
        s[2] = np.array([self.q.sum(dtype=np.uint64)], np.uint64).view(np.float64)[0]
        s[3] = (w * self.p[0]).sum(); s[4] = (w * self.p[1]).sum()
        s[5] = (w * np.sin(self.p[2])).sum(); s[6] = (w * np.cos(self.p[2])).sum()

    def stage_finish(self, sums5):
        self.global_sums = np.array(sums5)
