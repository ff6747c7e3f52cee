"""Particle set sharded over the GPUs of one node: one process per GPU, one engine per process.

Who runs the collectives of an update (three hosts over the same engine stages, bit-identical results):
  * backend "nccl", one rank per device (the production case): the ENGINE -- it holds its own RCCL communicator
    (mcl_comm_*, include/mcl_hip_engine.h) and `mcl_comm_update` is the whole update in one native call, the all-gather and the
    two all-reduces enqueued on the engine's stream between its kernels, one host wait.  torch.distributed then only carries
    the 128-byte rendezvous (`_make_native_comm`).
  * any backend, a GPU, MCL_DIST_NATIVE=0 or gloo (rehearsals with several ranks on one device): torch's collectives, ordered
    against the engine's stream by events (`_update_ordered`: mcl_stage_*_async, the reduced values stay in device memory,
    one host wait); MCL_DIST_SYNC=1: stage by stage, the host reading every value (rounds 1-3).
  * a CPU stand-in (tests): stage by stage over gloo.
What the exchange is, in every case:

The reference has no distributed code.  The update couples particles only through
  (1) resampling  — children of rank g are drawn from the GLOBAL weighted set.
                    LIST exchange (the usual case): after an update with many beams only a few per cent of the particles
                    keep a non-zero fixed-point weight, and a particle without weight is never selected (E6).  The scan of
                    every shard's weights leaves a compact list of the ones that carry weight (index, CDF value, record:
                    44 B each); ONE all-gather of those lists gives every rank the whole parent population -- no weight
                    gather, no pass over n_total elements, no index / request / reply round trips.  The list lengths and
                    weight totals of all ranks ride in the sums all-reduce of the previous update, so every rank sizes the
                    gather identically without a collective of its own.
                    DENSE exchange (some shard has no list: the first update after initialisation, flat weights): the
                    fixed-point weights are gathered (all-gather, 8 B per particle: every rank scans the same exact integer
                    CDF); each rank asks the engine which parents its children selected (`stage_resample_indices`), requests
                    the DISTINCT ones from their owners (all-to-alls: local indices out, 32-byte records back) and hands the
                    record table to `stage_motion_records`;
  (2) max log-weight — all-reduce(MAX) of one double;
  (3) normalisation / pose — all-reduce(SUM) of five doubles, plus three slots per rank that only that rank fills: the
                    length of its compact list and the two halves of its fixed-point weight total (exact in doubles), so
                    that the next update knows every shard's list and the global total.
Everything else (motion, ray cast, likelihood) is local to a shard.  Because the CDF is an exact
integer scan and the log-weights are exact fp64 sums, resample indices and weights are bit-identical
for any number of ranks.  (Several GPUs driven by ONE process do the same through `mcl_group_*` in the
library itself, with peer copies and peer pointers instead of collectives.)

Failure protocol (a rank that fails must not hang its peers): every update ends with the SUM all-reduce, and that vector carries one
more element, the ERROR WORD -- the number of ranks that failed in this update.  A rank whose engine call fails (or that finds its
state inconsistent, or MCL_DIST_FAIL="<rank>:<update>:<call prefix>" says so: the test switch) stops its local work but STILL
ENTERS EVERY COLLECTIVE of the update with the agreed sizes (`_local` wraps every engine call of an update; a rank that has failed
requests no parents, answers requests with whatever its buffers hold, contributes -inf and a zero vector) and raises the word.
After the exchange EVERY rank raises engine.ShardedUpdateError from the same update (`.local` tells whose failure it was); the
particle set must be set or initialised again (set_particles / reset()) before the next update.  What this cannot cover -- a rank
that dies or never arrives -- is bounded by the process group's timeout (torch.distributed.init_process_group(timeout=...)) in
this flow and by MCL_COMM_TIMEOUT_MS in the native one (include/mcl_hip_engine.h: mcl_comm_update).

`shard` is anything with the staging interface of engine.Engine (compact_list / export_compact / stage_resample_compact /
export_state / scan_weights / stage_resample_indices / stage_distinct_parents / export_records_at / stage_motion_records /
stage_rays / scalars / stage_weights / stage_finish); tests drive this class on CPU tensors over gloo with an oracle-backed
stand-in that lives under tests/.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from .engine import ShardedUpdateError


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
        self._exchange_bytes = dict(kind="none", weights_received=0, requests_sent=0, records_received=0, distinct_remote_parents=0,
                                   list_bytes_received=0, list_payload_bytes=0)
        # list exchange: every rank's list length (-1: none) and fixed-point total, known from the previous update's sums
        self.counts = None
        self.totals = None
        self.list_cap = max(4096, ((n // 4 + 63) // 64) * 64)             # the engine's own bound on a list (max_particles / 4)
        self.chunk_local = None                                           # grown on demand: one chunk / world chunks, uint8
        self.chunk_all = None
        self.pending_list = None                                          # (async all-gather of the lists, entries per chunk)
        self.use_lists = os.environ.get("MCL_DIST_NO_LISTS") != "1" and hasattr(shard, "export_compact")
        # the two small all-reduces of an update go through tensors made once (a fresh device tensor per update is an
        # allocation and a blocking copy each way: 0.1 ms of the update at 4M particles)
        # ([0]: the MAX exchange, [1:]: the SUM exchange with two more elements: the ray stage's overflow flag and sum w^2)
        # ... and, last, the error word of the failure protocol (ranks that failed in this update)
        self.red_dev = torch.zeros(1 + 5 + 3 * self.world + 2 + 1, dtype=torch.float64, device=device)
        self.red_host = torch.zeros(1 + 5 + 3 * self.world + 2 + 1, dtype=torch.float64)
        if device.type == "cuda":
            self.red_host = self.red_host.pin_memory()
        # device-ordered update (one host wait per update, the exchanged scalars never leave the device): needs the engine's
        # *_async stages; MCL_DIST_SYNC=1 keeps the stage-by-stage flow (A/B measurements, the CPU stand-in of the tests)
        self.device_ordered = (device.type == "cuda" and hasattr(shard, "stage_complete") and self.use_lists
                               and os.environ.get("MCL_DIST_SYNC") != "1")
        self._host_waits = 0                                               # of the last update (tests)
        # native exchange (one rank per DEVICE over RCCL): the engine holds its own RCCL communicator and runs the whole update
        # in one call, the collectives on its own stream (include/mcl_hip_engine.h: mcl_comm_*).  torch.distributed then only
        # carries the rendezvous and the first update's dense exchange.  Needs the nccl backend (a gloo rehearsal shares one
        # device between the ranks, which RCCL refuses); MCL_DIST_NATIVE=0 keeps the torch collectives.
        # adaptive resampling (Config.resample_neff_permille = r > 0, an option the reference does not have): the set is kept -- no
        # exchange, no resampling -- when the effective sample size of the WHOLE set after the previous update is >= r / 1000 of it
        self.neff_permille = int(getattr(getattr(shard, "cfg", None), "resample_neff_permille", 0)) if hasattr(shard, "stage_keep") else 0
        self.last_sw = self.last_sww = None                               # sum w and sum w^2 of the whole set after the last update
        self.kept_last = False
        self.native = False
        self.native_updates = 0                                           # consecutive updates the native call has run
        # (opt-in, MCL_DIST_NATIVE=1: no RCCL call of the engine's own communicator has run with more than one rank yet -- no
        #  multi-GPU box in any round -- so the default on a node is torch's collectives, whose RCCL use is the common one)
        if (self.device_ordered and hasattr(shard, "comm_update") and os.environ.get("MCL_DIST_NATIVE") == "1"
                and dist.get_backend(group) == "nccl"):
            self.native = self._make_native_comm()
        # failure protocol: what this rank found wrong in the current update (None: nothing), the update count, the test switch
        self._err = None
        self._pending_err = None                                          # a failure while preparing the NEXT update's exchange
        self.last_failed_ranks = 0                                        # the error word of the last update's SUM exchange (0: a good update)
        self.updates = 0
        self._fail_at = None
        spec = os.environ.get("MCL_DIST_FAIL")
        if spec:
            try:
                r, u, name = spec.split(":", 2)
                self._fail_at = (int(r), int(u), name)
            except ValueError:
                raise ValueError("MCL_DIST_FAIL must be <rank>:<update>:<engine call prefix>") from None

    @property
    def host_waits(self):
        """Stream synchronisations of the last update on this rank."""
        return self.shard.comm_stats()["host_waits"] if self.native_updates > 0 else self._host_waits

    @property
    def exchange_bytes(self):
        if self.native_updates > 0:
            st = self.shard.comm_stats()
            if st["dense"]:      # whole shards' weights and records (no request step: records_received counts every particle of the others)
                self._exchange_bytes.update(kind="dense", list_bytes_received=0, list_payload_bytes=0, weights_received=st["weights_received"],
                                            requests_sent=0, records_received=st["records_received"], distinct_remote_parents=0)
            else:
                self._exchange_bytes.update(kind="lists", list_bytes_received=st["list_bytes_received"], list_payload_bytes=st["list_payload_bytes"],
                                            weights_received=0, requests_sent=0, records_received=0, distinct_remote_parents=0)
        return self._exchange_bytes

    def _make_native_comm(self):
        """Every rank ends with the same answer: the communicator exists everywhere, or nowhere (torch collectives then)."""
        ok, why = self.shard.comm_available()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) != 1:
            if self.rank == 0:
                print(f"[dist] no RCCL for the engine on some rank ({why or 'another rank'}): torch collectives", file=sys.stderr)
            return False
        uid = torch.zeros(128, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            uid.copy_(torch.from_numpy(np.frombuffer(self.shard.comm_unique_id(), dtype=np.uint8).copy()))
        dist.broadcast(uid, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)

        def agreed(step):
            """Run `step` here; True when it succeeded on EVERY rank (the next collective step is only entered then)."""
            try:
                step()
                ok_here = 1
            except Exception as ex:                    # noqa: BLE001 -- any failure here means "not on this rank"
                print(f"[dist] rank {self.rank}: {ex}: torch collectives", file=sys.stderr)
                ok_here = 0
            flag.fill_(ok_here)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item()) == 1

        made = agreed(lambda: self.shard.comm_create(uid.cpu().numpy().tobytes(), self.world, self.rank))
        # the collectives of an update on known data, before a particle depends on them
        if made and agreed(self.shard.comm_selftest):
            return True
        try:
            self.shard.comm_destroy()
        except Exception:                              # noqa: BLE001
            pass
        return False

    # ---- failure protocol
    def _local(self, name, *args, default=None):
        """An engine call of the current update.  Skipped once this rank has failed; a failure is NOTED, not raised: the
        collectives of the update go on and the error word tells every rank."""
        if self._err is not None:
            return default
        f = self._fail_at
        if f is not None and f[0] == self.rank and f[1] == self.updates and name.startswith(f[2]):
            self._err = RuntimeError(f"injected failure before {name} (MCL_DIST_FAIL)")
            return default
        try:
            return getattr(self.shard, name)(*args)
        except Exception as ex:                        # noqa: BLE001 -- whatever the engine raised: this rank has failed
            self._err = ex
            return default

    def _void(self, failed_ranks):
        """Every rank arrives here from the same update: forget what the exchange knew and raise."""
        err, self._err = self._err, None
        self.last_failed_ranks = max(1, int(failed_ranks))
        self.pending_q = self.pending_list = None
        self.q_total = self.counts = self.totals = None
        self.last_sw = self.last_sww = None
        self.native_updates = 0
        if err is not None:
            raise ShardedUpdateError(f"rank {self.rank}: {err} [sharded update void on every rank]", getattr(err, "status", None), local=True) from err
        raise ShardedUpdateError(f"rank {self.rank}: {int(failed_ranks)} rank(s) of the sharded set reported a failure in this update: it is void on "
                                 "every rank (set or initialise the particles again)", -6, local=False)

    def _all_reduce_small(self, values, op):
        """values (a short float64 sequence) -> their reduction over the ranks, as a numpy array."""
        k = len(values)
        self.red_host[:k] = torch.as_tensor(np.asarray(values, np.float64))
        buf = self.red_dev[:k]
        buf.copy_(self.red_host[:k], non_blocking=True)
        dist.all_reduce(buf, op=op, group=self.group)
        self.red_host[:k].copy_(buf, non_blocking=True)
        self._sync()
        return self.red_host[:k].numpy().copy()

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
            self._host_waits += 1

    def reset(self):
        """Call after the shard's particle state was replaced from outside (set_particles / init_*)."""
        for pend in (self.pending_q, self.pending_list[0] if self.pending_list else None):
            if pend is not None:
                pend.wait()
                self._sync()
        self.pending_q = None
        self.pending_list = None
        self.q_total = None
        self.counts = self.totals = None
        self.last_sw = self.last_sww = None
        self._err = self._pending_err = None
        if self.native:
            self.shard.comm_set_lists([-1] * self.world, [0] * self.world)      # no lists: the next update is a dense one on every rank
            self.native_updates = 0

    # ---- list exchange
    def _lists_usable(self):
        return (self.use_lists and self.counts is not None and bool((self.counts >= 0).all()) and int(self.counts.max()) <= self.list_cap
                and int(self.totals.sum()) > 0)

    def _start_list_gather(self, async_op):
        """all-gather of the shards' compact lists as chunks of `entries` entries (the same on every rank: the longest list
        rounded up to 64); returns (work handle or None, entries)."""
        entries = max(64, (int(self.counts.max()) + 63) // 64 * 64)
        nbytes = 44 * entries
        if self.chunk_local is None or self.chunk_local.numel() < nbytes:
            self.chunk_local = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.chunk_all = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
        if int(self.counts[self.rank]) > 0:
            self._local("export_compact", self.chunk_local.data_ptr(), entries)
        self._sync()
        work = dist.all_gather_into_tensor(self.chunk_all[:nbytes * self.world], self.chunk_local[:nbytes], group=self.group, async_op=async_op)
        return (work if async_op else None), entries

    def _start_list_gather_ordered(self, async_op):
        """_start_list_gather without a host wait: the export runs on the engine's stream, the collective's stream waits for it."""
        entries = max(64, (int(self.counts.max()) + 63) // 64 * 64)
        nbytes = 44 * entries
        if self.chunk_local is None or self.chunk_local.numel() < nbytes:
            self.chunk_local = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.chunk_all = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
        ts = torch.cuda.current_stream(self.device).cuda_stream
        self._local("stream_wait_external", ts)      # (the buffers may be fresh, or still read by the previous update's collective)
        if int(self.counts[self.rank]) > 0:
            self._local("export_compact_async", self.chunk_local.data_ptr(), entries)
        self._local("external_wait_stream", ts)
        work = dist.all_gather_into_tensor(self.chunk_all[:nbytes * self.world], self.chunk_local[:nbytes], group=self.group, async_op=async_op)
        return (work if async_op else None), entries

    def _update_ordered(self, action, obs, keep=False):
        """One update with the lists known (every update but the first): everything is enqueued, the two small exchanges read
        and write device memory, the host waits once -- for the summed vector.  keep: adaptive resampling kept the set (no lists)."""
        s, world = self.shard, self.world
        ts = torch.cuda.current_stream(self.device).cuda_stream
        if keep:
            if self.pending_list is not None and self.pending_list[0] is not None:
                self.pending_list[0].wait()
            self.pending_list = None
            self._exchange_bytes.update(kind="none", list_bytes_received=0, list_payload_bytes=0, weights_received=0,
                                        requests_sent=0, records_received=0, distinct_remote_parents=0)
            self._local("stream_wait_external", ts)
            self._local("stage_keep", self.rank * self.n, self.n_total, action)
            return self._ordered_after_children(obs, ts)
        if self.pending_list is not None:
            work, entries = self.pending_list                             # issued at the end of the previous update
            self.pending_list = None
            if work is not None:
                work.wait()                                               # (the current stream waits, not the host)
        else:
            _, entries = self._start_list_gather_ordered(False)
        listed = int(self.counts.sum())
        self._exchange_bytes.update(kind="lists", list_bytes_received=44 * entries * (world - 1),
                                   list_payload_bytes=44 * (listed - int(self.counts[self.rank])), weights_received=0,
                                   requests_sent=0, records_received=0, distinct_remote_parents=0)
        self._local("stream_wait_external", ts)
        self._local("stage_resample_compact_async", self.chunk_all.data_ptr(), world, entries, self.counts, self.totals, self.n, self.rank,
                    self.rank * self.n, self.n_total, action)
        return self._ordered_after_children(obs, ts)

    def _ordered_after_children(self, obs, ts):
        s, world = self.shard, self.world
        k = 5 + 3 * world + 2
        self._local("stage_rays_async", obs, self.red_dev.data_ptr())    # local max log-weight -> red_dev[0]
        self._local("external_wait_stream", ts)
        if self._err is not None:
            self.red_dev[:1].fill_(float("-inf"))
        dist.all_reduce(self.red_dev[:1], op=dist.ReduceOp.MAX, group=self.group)
        self._local("stream_wait_external", ts)
        vec = self.red_dev[1:1 + k + 1]                                   # the summed vector and, behind it, the error word
        self._local("stage_weights_async", self.red_dev.data_ptr(), vec.data_ptr(), world, self.rank)
        self._local("external_wait_stream", ts)
        if self._err is not None:
            vec.zero_()
        vec[k:].fill_(0.0 if self._err is None else 1.0)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
        self.red_host[:1 + k + 1].copy_(self.red_dev[:1 + k + 1], non_blocking=True)
        self._sync()                                                      # THE host wait of the update
        gs = self.red_host[1:1 + k].numpy().copy()
        failed = float(self.red_host[1 + k])
        if failed != 0.0 or self._err is not None:
            self._void(failed)
        s.stage_complete(gs[:5])
        if gs[-2] != 0.0:
            # some shard's fix-up lists overflowed (debug_force_exact at size, a pathological map): every rank runs the ray stage
            # and the two exchanges once more, stage by stage (the synchronous ray stage falls back to the self-contained kernel)
            return None
        return np.concatenate([gs[:-2], gs[-1:]])                        # [5 sums | 3 per shard | sum w^2]

    def _distinct(self):
        """self.parent (global indices) -> (distinct parents ascending = grouped by owner, position of every child's parent
        among them): a bitmap over the global indices and its popcount prefix inside the engine (passes over
        n_total / 32 words, not n_total elements)."""
        k = self._local("stage_distinct_parents", self.parent.data_ptr(), self.n, self.n_total, self.uniq_buf.data_ptr(), self.slot_buf.data_ptr(),
                        default=0)              # (a rank that has failed requests no parents)
        return self.uniq_buf[:k], self.slot_buf

    def _records_at(self, local_idx, out):
        """Records of this shard's particles `local_idx` (int64 tensor) -> `out` ((k, 4) float64 tensor)."""
        if local_idx.numel() == 0:
            return out
        self._sync()                 # the engine has its own stream: the indices (tensor ops / a collective) must be complete
        self._local("export_records_at", local_idx.data_ptr(), int(local_idx.numel()), out.data_ptr())
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
            self._exchange_bytes.update(requests_sent=0, records_received=0, distinct_remote_parents=0)
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
        self._exchange_bytes.update(requests_sent=8 * remote, records_received=32 * remote, distinct_remote_parents=remote)
        return table, inv

    def _resample_from_lists(self, action):
        if self.pending_list is not None:
            work, entries = self.pending_list                             # issued at the end of the previous update
            self.pending_list = None
            if work is not None:
                work.wait()
        else:
            _, entries = self._start_list_gather(False)
        self._sync()
        listed = int(self.counts.sum())
        # what crosses the links is the padded chunk of every other rank (the all-gather moves `entries` entries per rank: the
        # longest list rounded up to 64); the entries that carry data are reported beside it
        self._exchange_bytes.update(kind="lists", list_bytes_received=44 * entries * (self.world - 1),
                                   list_payload_bytes=44 * (listed - int(self.counts[self.rank])), weights_received=0,
                                   requests_sent=0, records_received=0, distinct_remote_parents=0)
        self._local("stage_resample_compact", self.chunk_all.data_ptr(), self.world, entries, self.counts, self.totals, self.n, self.rank,
                    self.rank * self.n, self.n_total, action)

    def _resample_dense(self, action):
        s = self.shard
        if self.pending_q is None:
            self._local("export_state", 0, 0, 0, self.loc_q.data_ptr())
            dist.all_gather_into_tensor(self.glob_q, self.loc_q, group=self.group)
        else:
            self.pending_q.wait()                                        # issued at the end of the previous update
            self.pending_q = None
        self._sync()
        self._exchange_bytes.update(kind="dense", weights_received=8 * self.n * (self.world - 1), list_bytes_received=0, list_payload_bytes=0)
        self._local("scan_weights", self.glob_q.data_ptr(), self.glob_cdf.data_ptr(), self.n_total, 0)
        q_total = self.q_total if self.q_total is not None else int(self.glob_cdf[-1].item()) & 0xFFFFFFFFFFFFFFFF
        self._local("stage_resample_indices", self.glob_cdf.data_ptr(), self.n_total, q_total, self.rank * self.n, self.n_total, self.parent.data_ptr())
        table, slot = self._fetch_parents()
        self._sync()
        self._local("stage_motion_records", table.data_ptr(), int(table.shape[0]), slot.data_ptr(), self.rank * self.n, self.n_total, action)

    def update(self, action, obs):
        s = self.shard
        self._host_waits = 0
        self.updates += 1
        self.last_failed_ranks = 0
        self._err, self._pending_err = self._pending_err, None            # (a failure while preparing this update's exchange is this update's)
        gs = None
        keep = False
        if self.neff_permille > 0 and self.last_sw is not None and not self.native:
            keep = self.last_sww > 0.0 and self.last_sw * self.last_sw >= (self.neff_permille / 1000.0) * float(self.n_total) * self.last_sww
        self.kept_last = keep
        # (1) exchange for resampling + the children
        if self.native:
            # the whole update in one native call -- lists, or the dense exchange when there are none (first update) -- on the
            # engine's stream.  A failure on any rank raises ShardedUpdateError on every rank (the engine's own protocol).
            try:
                pose = s.comm_update(action, obs)
            except ShardedUpdateError:
                self.native_updates = 0
                self.q_total = self.counts = self.totals = None
                raise
            self.pose = pose
            self.native_updates += 1
            self.q_total = self.counts = self.totals = None               # (the communicator keeps them now)
            self.kept_last = bool(s.comm_stats()["kept"]) if self.neff_permille > 0 else False
            return pose
        elif self.device_ordered and (keep or self._lists_usable()):
            gs = self._update_ordered(action, obs, keep)                 # the whole update; None: once more from the ray stage on
        elif keep:
            if self.pending_list is not None and self.pending_list[0] is not None:
                self.pending_list[0].wait()
            if self.pending_q is not None:
                self.pending_q.wait()
            self.pending_list = self.pending_q = None
            self._sync()
            self._exchange_bytes.update(kind="none", list_bytes_received=0, list_payload_bytes=0, weights_received=0,
                                        requests_sent=0, records_received=0, distinct_remote_parents=0)
            self._local("stage_keep", self.rank * self.n, self.n_total, action)
        elif self._lists_usable():
            self._resample_from_lists(action)
        else:
            if self.pending_list is not None and self.pending_list[0] is not None:
                self.pending_list[0].wait()
            self.pending_list = None
            self._resample_dense(action)
        if gs is None:
            gs = self._stages_after_children(obs)
        per = gs[5:5 + 3 * self.world].reshape(self.world, 3)
        self.last_sw, self.last_sww = float(gs[0]), float(gs[-1])          # (adaptive resampling: N_eff of the whole set)
        self.counts = per[:, 0].astype(np.int64) - 1
        self.totals = np.array([(int(a) + (int(b) << 32)) & 0xFFFFFFFFFFFFFFFF for a, b in per[:, 1:]], dtype=np.uint64)   # exact: halves < 2^32
        self.q_total = int(sum(int(t) for t in self.totals)) & 0xFFFFFFFFFFFFFFFF
        gs = gs[:5]
        s.stage_finish(gs)
        next_keeps = (self.neff_permille > 0 and self.last_sww > 0.0
                      and self.last_sw * self.last_sw >= (self.neff_permille / 1000.0) * float(self.n_total) * self.last_sww)
        if self.overlap and not next_keeps:
            # this update's weights are final: start the exchange of the next update now, beside the host work between updates
            # (a failure of these engine calls belongs to the NEXT update: its exchange is entered all the same, then it is void)
            if self._lists_usable():
                self.pending_list = (self._start_list_gather_ordered if self.device_ordered else self._start_list_gather)(True)
            else:
                self._local("export_state", 0, 0, 0, self.loc_q.data_ptr())
                self.pending_q = dist.all_gather_into_tensor(self.glob_q, self.loc_q, group=self.group, async_op=True)
            self._pending_err, self._err = self._err, None
        k = 1.0 / gs[0] if gs[0] > 0 else 1.0
        self.pose = np.array([gs[1] * k, gs[2] * k, np.arctan2(gs[3] * k, gs[4] * k)])
        return self.pose

    def _stages_after_children(self, obs):
        """Ray stage, MAX exchange, weights, SUM exchange -- stage by stage, the host reading each value (first update, dense
        exchange, the CPU stand-in, the redo after an overflow).  Returns the summed vector."""
        s = self.shard
        self._local("stage_rays", obs)
        # (2) global max log-weight
        read = "host_scalars" if hasattr(s, "host_scalars") else "scalars"      # the stage calls already read SCALARS back
        sc = self._local(read)
        local_max = float(sc[0]) if self._err is None else float("-inf")
        self._local("stage_weights", float(self._all_reduce_small([local_max], dist.ReduceOp.MAX)[0]))
        # (3) global sums: sum w, sum wx, sum wy, sum w sin, sum w cos; per rank (filled by that rank only): list length + 1
        # (0: no list) and the two halves of its fixed-point weight total; sum w^2; the error word
        sc = self._local(read)
        cl = self._local("compact_list") if self.use_lists else None
        vec = np.zeros(5 + 3 * self.world + 2)
        if self._err is None:
            ql = int(np.float64(sc[2]).view(np.uint64))                  # this shard's fixed-point weight total
            n_list = cl[0] if cl is not None else -1
            vec[:5] = (sc[1], sc[3], sc[4], sc[5], sc[6])
            vec[5 + 3 * self.rank: 8 + 3 * self.rank] = (float(n_list + 1), float(ql & 0xFFFFFFFF), float(ql >> 32))
            vec[-2] = sc[7] if len(sc) > 7 else 0.0                      # sum w^2 (adaptive resampling)
        else:
            vec[-1] = 1.0
        out = self._all_reduce_small(vec, dist.ReduceOp.SUM)
        if out[-1] != 0.0 or self._err is not None:
            self._void(out[-1])
        return out[:-1]

    def set_particles(self, xyz_colmajor, weights):
        """Host-supplied particles for this shard (any non-negative weights): every shard quantises its weights against the
        maximum of the WHOLE set, found with one all-reduce."""
        w = np.ascontiguousarray(weights, np.float64)
        mx = torch.tensor([float(w.max())], dtype=torch.float64, device=self.device)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        self.shard.set_particles_shard(xyz_colmajor, w, float(mx.item()))
        self.reset()

    def expected_pose(self):
        return self.pose

    def effective_sample_size(self):
        """(N_eff = (sum w)^2 / sum w^2 of the WHOLE set after the last update, whether that update resampled) -- what
        Engine.effective_sample_size reports for one engine; None before the first update."""
        if self.native and self.native_updates > 0:
            v = self.shard.comm_vector()
            sw, sww = float(v[0]), float(v[-1])
        elif self.last_sw is not None:
            sw, sww = self.last_sw, self.last_sww
        else:
            return None
        return (sw * sw / sww if sww > 0.0 else 0.0), not self.kept_last
