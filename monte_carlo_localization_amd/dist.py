"""Particle set sharded over the GPUs of one node: one process per GPU, one engine per process,
torch.distributed (backend "nccl" == RCCL over xGMI) for the exchange steps of an update.

The reference has no distributed code.  The update couples particles only through
  (1) resampling  — children of rank g are drawn from the GLOBAL weighted set.  Only the fixed-point weights are
                    gathered (all-gather, 8 B per particle: every rank scans the same exact integer CDF); each rank then
                    asks the engine which parents its children selected (`stage_resample_indices`), requests the DISTINCT
                    ones from their owners (two all-to-alls: 4-byte local indices out, 32-byte records back) and hands the
                    compact record table to `stage_motion_records`.  With the peaked weights of a converged filter a few
                    thousand parents serve millions of children, so almost nothing crosses a link; the worst case
                    (uniform weights) moves what the former wholesale all-gather of the records always moved;
  (2) max log-weight — all-reduce(MAX) of one double;
  (3) normalisation / pose — all-reduce(SUM) of five doubles (+ the two halves of the local
                    fixed-point weight total, so that the next update knows the global total
                    without reading it back from the device).
Everything else (motion, ray cast, likelihood) is local to a shard.  Because the CDF is an exact
integer scan and the log-weights are exact fp64 sums, resample indices and weights are bit-identical
for any number of ranks.  (Several GPUs driven by ONE process do the same through `mcl_group_*` in the
library itself, with peer copies and peer pointers instead of collectives.)

`shard` is anything with the staging interface of engine.Engine (export_state / scan_weights /
stage_resample_indices / stage_distinct_parents / export_records_at / stage_motion_records / stage_rays / scalars /
stage_weights / stage_finish); tests drive this class on CPU tensors over gloo with an oracle-backed stand-in that lives
under tests/.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


class ShardedFilter:
    """overlap=True: the fixed-point weights are all-gathered for the NEXT update as soon as this update has produced
    them, beside the sums all-reduce and the host work between updates."""

    def __init__(self, shard, n_local: int, device: torch.device, group=None, overlap: bool = True, reserved_cus: int = 0):
        self.shard = shard
        self.n = int(n_local)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        # a single rank has nothing to overlap; MCL_FORCE_OVERLAP=1 keeps the path on for rehearsals
        self.overlap = bool(overlap) and (self.world > 1 or os.environ.get("MCL_FORCE_OVERLAP") == "1")
        if reserved_cus and hasattr(shard, "set_reserved_cus"):
            shard.set_reserved_cus(reserved_cus)
        f64, i64, i32 = torch.float64, torch.int64, torch.int32
        n, nt = self.n, self.n * self.world
        self.n_total = nt
        self.loc_q = torch.empty(n, dtype=i64, device=device)           # uint64 bits
        self.q_total = None                                              # global fixed-point weight total, once known
        self.pending_q = None                                            # async gather of the weights issued by the previous update
        self.glob_q = torch.empty(nt, dtype=i64, device=device)
        self.glob_cdf = torch.empty(nt, dtype=i64, device=device)
        self.parent = torch.empty(n, dtype=i32, device=device)          # global parent index of every local child
        self.uniq_buf = torch.empty(n, dtype=i64, device=device)        # distinct parents of the local children (ascending)
        self.slot_buf = torch.empty(n, dtype=i32, device=device)        # every child's position among them
        self.pose = np.zeros(3)
        self.exchange_bytes = dict(weights_received=0, requests_sent=0, records_received=0, distinct_remote_parents=0)

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def reset(self):
        """Call after the shard's particle state was replaced from outside (set_particles / init_*)."""
        if self.pending_q is not None:
            self.pending_q.wait()
            self._sync()
        self.pending_q = None
        self.q_total = None

    def _distinct(self):
        """self.parent (global indices) -> (distinct parents ascending = grouped by owner, position of every child's parent
        among them): a bitmap over the global indices and its popcount prefix inside the engine (passes over
        n_total / 32 words, not n_total elements)."""
        k = self.shard.stage_distinct_parents(self.parent.data_ptr(), self.n, self.n_total, self.uniq_buf.data_ptr(), self.slot_buf.data_ptr())
        return self.uniq_buf[:k], self.slot_buf

    def _records_at(self, local_idx, out):
        """Records of this shard's particles `local_idx` (int64 tensor) -> `out` ((k, 4) float64 tensor)."""
        if local_idx.numel() == 0:
            return out
        self._sync()                 # the engine has its own stream: the indices (tensor ops / a collective) must be complete
        self.shard.export_records_at(local_idx.data_ptr(), int(local_idx.numel()), out.data_ptr())
        return out

    def _fetch_parents(self):
        """self.parent (global indices) -> (compact record table, position of every child's parent in it).  Distinct
        parents only; the ones this rank owns are read locally, the others requested from their owners."""
        n, world = self.n, self.world
        uniq, inv = self._distinct()
        k = int(uniq.numel())
        table = torch.empty((k, 4), dtype=torch.float64, device=self.device)
        # owners' shares of the ascending list: world + 1 boundaries
        bounds = torch.searchsorted(uniq, torch.arange(world + 1, dtype=torch.int64, device=self.device) * n)
        sc = (bounds[1:] - bounds[:-1]).tolist()
        local = (uniq - torch.div(uniq, n, rounding_mode="floor") * n).contiguous()
        if world == 1:
            self._records_at(local, table)
            self.exchange_bytes.update(requests_sent=0, records_received=0, distinct_remote_parents=0)
            return table, inv
        send_counts = torch.tensor(sc, dtype=torch.int64, device=self.device)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        rcv = recv_counts.tolist()
        req_in = torch.empty(int(sum(rcv)), dtype=torch.int64, device=self.device)
        dist.all_to_all_single(req_in, local, output_split_sizes=rcv, input_split_sizes=sc, group=self.group)
        reply = torch.empty((int(req_in.numel()), 4), dtype=torch.float64, device=self.device)
        self._records_at(req_in, reply)                                  # the records the other ranks asked this one for
        self._sync()
        dist.all_to_all_single(table, reply, output_split_sizes=sc, input_split_sizes=rcv, group=self.group)
        remote = k - sc[self.rank]
        self.exchange_bytes.update(requests_sent=8 * remote, records_received=32 * remote, distinct_remote_parents=remote)
        return table, inv

    def update(self, action, obs):
        s = self.shard
        # (1) exchange for resampling: weights everywhere, then only the selected parents
        if self.pending_q is None:
            s.export_state(0, 0, 0, self.loc_q.data_ptr())
            dist.all_gather_into_tensor(self.glob_q, self.loc_q, group=self.group)
        else:
            self.pending_q.wait()                                        # issued at the end of the previous update
            self.pending_q = None
        self._sync()
        self.exchange_bytes["weights_received"] = 8 * self.n * (self.world - 1)
        s.scan_weights(self.glob_q.data_ptr(), self.glob_cdf.data_ptr(), self.n_total, 0)
        q_total = self.q_total if self.q_total is not None else int(self.glob_cdf[-1].item()) & 0xFFFFFFFFFFFFFFFF
        s.stage_resample_indices(self.glob_cdf.data_ptr(), self.n_total, q_total, self.rank * self.n, self.n_total, self.parent.data_ptr())
        table, slot = self._fetch_parents()
        self._sync()
        s.stage_motion_records(table.data_ptr(), int(table.shape[0]), slot.data_ptr(), self.rank * self.n, self.n_total, action)
        s.stage_rays(obs)
        # (2) global max log-weight
        read = getattr(s, "host_scalars", s.scalars)                      # the stage calls already read SCALARS back
        mx = torch.tensor([read()[0]], dtype=torch.float64, device=self.device)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        s.stage_weights(float(mx.item()))
        if self.overlap:
            # this update's fixed-point weights are final: start gathering them for the next update now, beside the
            # sums all-reduce and the host work between updates
            s.export_state(0, 0, 0, self.loc_q.data_ptr())
            self.pending_q = dist.all_gather_into_tensor(self.glob_q, self.loc_q, group=self.group, async_op=True)
        # (3) global sums: sum w, sum wx, sum wy, sum w sin, sum w cos
        sc = read()
        ql = int(np.float64(sc[2]).view(np.uint64))                      # this shard's fixed-point weight total
        sums = torch.tensor([sc[1], sc[3], sc[4], sc[5], sc[6], float(ql & 0xFFFFFFFF), float(ql >> 32)],
                            dtype=torch.float64, device=self.device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        gs = sums.cpu().numpy()
        self.q_total = (int(gs[5]) + (int(gs[6]) << 32)) & 0xFFFFFFFFFFFFFFFF     # exact: both halves stay below 2^53
        gs = gs[:5]
        s.stage_finish(gs)
        k = 1.0 / gs[0] if gs[0] > 0 else 1.0
        self.pose = np.array([gs[1] * k, gs[2] * k, np.arctan2(gs[3] * k, gs[4] * k)])
        return self.pose

    def expected_pose(self):
        return self.pose
