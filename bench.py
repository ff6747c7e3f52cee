#!/usr/bin/env python3
"""bench.py — MCL updates/s as particle*beam/s on synthetic scans (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full update (resample -> motion -> ray cast + likelihood -> normalise -> expected
pose) of the 4M-particle x 1081-beam Spielberg workload (BASELINE.json configs[2]; SURVEY.md §8(d)
inputs), per GPU.  N > 1 is launched by torch.distributed.run, one rank per GPU; particles are
sharded 4M per GPU (weak scaling, configs[4] at N=8) with RCCL per update: all-gather of the
fixed-point weights, all-to-all fetch of the DISTINCT selected parents, two small all-reduces
(monte_carlo_localization_amd/dist.py).  Inputs are resident in HBM before the timed region; the
per-update host->device traffic is the 1081-float scan and the 3-double action only.

Rank 0 prints ONE JSON line.

`roofline` names the bound that applies to the dominant kernel (k_rays_sweep): VALU issue (a wave64 VALU instruction occupies
its SIMD for one quad-cycle; 1024 SIMDs x 2.4 GHz).  `roofline.frac` is the USEFUL-WORK fraction -- the instructions the rays of
the launch need with every lane busy (per ray and per probe trip, trips counted live) over the live kernel time (HIP events on
the engine's stream); `roofline.valu.frac` is the ISSUE fraction, from the kernel's measured instruction count
(profiles/rNN_roofline_inputs.json, which tools/roofline_inputs.py writes from the committed rocprofv3 PMC passes).
`roofline.algorithmic` keeps SURVEY.md 8(d)'s figure (per ray S-bar one-byte grid probes as the reference reads them, cpp:642, +
one 4-byte table entry, cpp:576; per particle 32 B) priced against the HBM peak: it exceeds 1 because the kernel examines a
tenth of those samples (same results) and is NOT a bound.
`cpu_baseline` times the CPU oracle's as-reference step (same materialised arrays and the same
`omp parallel for schedule(dynamic)` ray loop as cpp:593) on this box's host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_PER_GPU = 4 * 1024 * 1024
ACTION = (0.05, 0.0, 0.01)
TRUE_POSE = (0.0, 0.0, 0.0)


def _latest_roofline_inputs():
    """profiles/rNN_roofline_inputs.json of the latest round that committed one (tools/profile_round.sh writes it)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_roofline_inputs.json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r02_roofline_inputs.json")


ROOFLINE_INPUTS = _latest_roofline_inputs()


def host_cpu():
    """CPU model string and the cores this process may run on (affinity), for the cpu_baseline block."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, cores, os.cpu_count() or cores


def cpu_baseline(m, ang, scan, true_pose=TRUE_POSE, budget_s=20.0, big=False):
    """Oracle (kind 'port'): BASELINE config #1 = 4000 particles x 1081 beams, the as-reference step (same materialised
    arrays, same `omp parallel for schedule(dynamic)` ray loop as cpp:593).  The figure of record is the ONE-thread one:
    the reference's chunk-1 dynamic schedule does not scale, so more threads are reported beside it, never
    instead of it.  Also: the reference's own six-stage split (TimingStats order, utils.hpp:51-57) and, with `big`, one
    update at 262 144 x 1081 (~11 GB of temporaries, about a minute)."""
    from oracle import oracle as orc
    om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
    n = 4000
    T = orc.sensor_table(om.max_range_px)
    s = orc.RefStream(42)
    p, w = orc.init_particles_pose(s, true_pose, n)
    model, cores, logical = host_cpu()
    out, stages = {}, {}
    for label, thr, share in (("one", 1, 0.4), ("three", min(3, cores), 0.2), ("all", cores, 0.4)):   # 3 = stock num_threads (yaml:40)
        orc.omp_threads(thr)
        pp, ww = p.copy(), w.copy()
        times, split, t_start = [], [], time.perf_counter()
        while True:
            u, nrm = s.uniforms(n), s.normals(3 * n).reshape(n, 3)
            t0 = time.perf_counter()
            r = orc.mcl_step(om, pp, ww, ACTION, ang, scan, T, u, nrm, use_omp=True, want_steps=False)
            times.append(time.perf_counter() - t0)
            split.append(r["timing_ms"])
            pp, ww = r["particles"], np.full(n, 1.0 / n)   # the reference product underflows at 1081 beams (D4)
            if len(times) >= 2 and time.perf_counter() - t_start > budget_s * share:
                break
        out[label] = (n * ang.size / float(np.median(times)), len(times), thr)
        stages[label] = [float(v) for v in np.median(np.array(split), axis=0)]
    base = {"value": out["one"][0], "unit": "particle*beam/s", "cores": 1, "kind": "port",
            "sample": f"{out['one'][1]} updates of 4000 particles x {ang.size} beams (BASELINE config #1), Spielberg_map, "
                      f"tracking-regime cloud, one thread (figure of record: the reference's omp schedule(dynamic) chunk-1 ray "
                      f"loop, cpp:593, does not scale)",
            "cpu_model": model, "cores_available": cores, "logical_cpus": logical,
            "three_thread_value": out["three"][0], "all_cores_value": out["all"][0], "all_cores_threads": out["all"][2],
            "stage_ms_one_thread": dict(zip(("resample", "motion", "query_prep", "ray_cast", "table_eval", "total"), stages["one"])),
            "stage_ms_all_cores": dict(zip(("resample", "motion", "query_prep", "ray_cast", "table_eval", "total"), stages["all"]))}
    if big:
        nb = 262144
        orc.omp_threads(1)
        pb = np.tile(p, (1, nb // n + 1))[:, :nb].copy()
        u, nrm = s.uniforms(nb), s.normals(3 * nb).reshape(nb, 3)
        t0 = time.perf_counter()
        r = orc.mcl_step(om, pb, np.full(nb, 1.0 / nb), ACTION, ang, scan, T, u, nrm, use_omp=True, want_steps=False)
        dt = time.perf_counter() - t0
        base["config_262144"] = {"value": nb * ang.size / dt, "seconds": dt, "threads": 1,
                                 "stage_ms": dict(zip(("resample", "motion", "query_prep", "ray_cast", "table_eval", "total"),
                                                      [float(v) for v in r["timing_ms"]]))}
    orc.omp_threads(cores)
    return base


MAX_CLOCK_GHZ = 2.4            # MI355X_MICROARCH.md chip table: max clock 2400 MHz
SIMDS = 1024                   # 256 CUs x 4 SIMDs


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


# VALU instructions of k_rays_sweep's walk per ray outside the trips / per trip, by the form that ran (csrc/mcl_rays_sweep.h;
# tools/roofline_inputs.py checks the trip against the shipped binary):
#   fetched directions (TAB): 4 FMA + index add | table: add, mad, add                                          =  8   per ray
#   turned directions (REC):  4 (integers) + 2 (three-term step) + 1/4 index add | table: add, mad, 1/4 add     =  8.5 per ray
#   two rays per lane (REC + PAIRS): 8 + 4 + index add | 2 add, 2 mad, add per PAIR                              =  9   per ray
#   hybrid (REC in LDS windows, the global fields beyond): REC + the compare for "left the window"               =  9.5 per ray
#   trip: 2 v_mad_u64_u32, address, v_min3_u32, v_sub_co_u32                                                    =  5   per trip
WALK_VALU = {"tab": (8.0, 5.0), "rec": (8.5, 5.0), "pairs": (9.0, 5.0), "hyb": (9.5, 5.0)}


def roofline_block(kernel_name, k_ms, n, B, sbar, profiled_workload=True, trips_live=None, variant=None):
    """The bound of the dominant kernel: VALU issue.  A wave64 VALU instruction occupies its SIMD for one quad-cycle whatever its
    kind (profiles/r05_pmc_sq2.csv: SQ_ACTIVE_INST_VALU, in quad-cycles, equals SQ_INSTS_VALU); the chip issues on 1024 SIMDs at
    up to 2.4 GHz.

    `frac` is the USEFUL-WORK fraction: what the rays of this launch need with all 64 lanes busy -- per ray the instructions of the
    walk outside the trips and five per probe trip, trips per ray counted live by an untimed update -- priced at 4 cycles each
    against 1024 SIMDs x 2.4 GHz, over the live kernel time.  The numerator does not grow with the kernel's own instruction count:
    idle lanes in the lock-step trips (a wave makes 4.5 trips for a per-lane mean of 3.4), per-chunk set-up, window loads,
    barriers and a clock below the maximum all lower it.
    `valu.frac` is the ISSUE fraction: 4 cycles x the kernel's measured VALU instruction count (SQ_INSTS_VALU, a profile constant
    of the default workload) over the same denominator -- how close the kernel as written is to the issue limit (the rest is waves
    parked in s_waitcnt: SQ_WAIT_ANY).
    `algorithmic` keeps SURVEY.md 8(d)'s bytes-over-HBM figure; it is not a bound (see the module docstring)."""
    alg_bytes = n * B * (sbar * 1.0 + 4.0) + n * 32.0       # per launch (one GPU's shard), SURVEY 8(d)
    alg = {"bytes_per_launch": alg_bytes, "s_bar_probes_per_ray": sbar, "achieved": alg_bytes / (k_ms * 1e-3) / 1e9,
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "note": "SURVEY 8(d) algorithmic bytes / kernel time / 8 TB/s: not a bound (the kernel skips ~90 % of the priced samples)"}
    block = {"bound": "valu", "kernel": kernel_name, "kernel_ms": k_ms, "achieved": None, "peak": SIMDS * MAX_CLOCK_GHZ,
             "unit": "G SIMD issue cycles/s", "frac": None, "algorithmic": alg, "traffic": None, "useful": None, "valu": None}
    if variant is not None:
        block["kernel_form"] = variant
    if trips_live is not None and kernel_name == "k_rays_sweep":
        v = variant or {}
        form = "tab" if not v.get("turned_directions") else "hyb" if v.get("hybrid") else "pairs" if v.get("pairs") else "rec"
        per_ray, per_trip = WALK_VALU[form]
        insts_per_ray = per_ray + per_trip * trips_live
        useful_ms = n * B / 64.0 * insts_per_ray * 4.0 / (SIMDS * MAX_CLOCK_GHZ * 1e6)
        block["useful"] = {"frac": useful_ms / k_ms, "floor_ms": useful_ms, "rays": n * B, "probe_trips_per_ray": trips_live,
                           "valu_per_ray": insts_per_ray, "issue_cycles_per_ray": 4.0 * insts_per_ray, "form": form,
                           "note": f"rays x ({per_ray} + {per_trip} x live trips per ray) VALU x 4 cycles / 64 lanes / (1024 SIMDs x 2.4 GHz) / live kernel ms"}
        block["frac"] = useful_ms / k_ms
        block["achieved"] = n * B / 64.0 * insts_per_ray * 4.0 / (k_ms * 1e-3) / 1e9
    inp = None
    if os.path.exists(ROOFLINE_INPUTS):
        try:
            inp = json.load(open(ROOFLINE_INPUTS))
        except Exception:
            inp = None
    # the committed instruction counts are those of the DEFAULT workload (Spielberg_map, tracking-regime cloud): another map
    # or cloud executes a different number of probe trips, so nothing is priced for it
    if inp and profiled_workload and inp.get("kernel") == kernel_name and inp.get("particles") == n and inp.get("beams") == B:
        I, T = inp["valu_insts_per_launch"], inp["lds_insts_per_launch"]
        cycles = 4.0 * I * inp.get("quad_cycles_per_valu_inst", 1.0)
        floor_ms = cycles / (SIMDS * MAX_CLOCK_GHZ * 1e6)
        block["valu"] = {"insts_per_launch": I, "lds_insts_per_launch": T, "issue_cycles_per_launch": cycles, "simds": SIMDS,
                         "clock_ghz": MAX_CLOCK_GHZ, "floor_ms": floor_ms, "frac": floor_ms / k_ms,
                         "quad_cycles_per_valu_inst": inp.get("quad_cycles_per_valu_inst"),
                         "wave_cycles_parked_frac": inp.get("wait_any_frac"), "clock_ghz_while_profiled": inp.get("clock_ghz"),
                         "kernel_ms_while_profiled": inp.get("kernel_ms_while_profiled"),
                         "source": os.path.relpath(ROOFLINE_INPUTS, ROOT), "source_sha256_16": _sha16(ROOFLINE_INPUTS),
                         "note": "insts_per_launch is a PROFILE CONSTANT (rocprofv3 PMC passes of this workload, committed under "
                                 "profiles/); only kernel_ms is measured by this run"}
        block["traffic"] = inp.get("hbm_bytes_per_launch")
        block["traffic_source"] = inp.get("hbm_bytes_source")
    else:
        block["note"] = "no profile inputs for this kernel / size / workload under profiles/: the issue fraction is not priced"
    return block


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--beam-step", type=int, default=1, help="keep every k-th of the 1081 beams (angle_step, cpp:307-310)")
    ap.add_argument("--map", choices=["spielberg", "levine", "fine025"], default="spielberg",
                    help="levine = the synthetic 2049x2049 @0.05 stand-in (maps/levine.pgm is absent from the reference); fine025 = "
                         "a synthetic 0.025 m map (MAX_RANGE_PX = 479: cpp:195 puts no bound on it), the long-range case")
    ap.add_argument("--regime", choices=["tracking", "global"], default="tracking")
    ap.add_argument("--resample", choices=["multinomial", "systematic"], default="multinomial")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-256k", action="store_true",
                    help="also time one oracle update at 262 144 x 1081 on one thread (SURVEY 8(d); about a minute, 11 GB)")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="sharded runs: gather the exchange synchronously")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the sharded (torch.distributed/RCCL) path even with one rank (rehearsal)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of the sharded path; gloo + --one-device rehearses --gpus N on a one-GPU box")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    return ap.parse_args(argv)


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks (one per GPU) as children BEFORE
    anything here has touched the GPU, stays GPU-free itself, hands rank 0's JSON line through and fails if any rank
    fails.  (Under torch.distributed.run the ranks arrive with WORLD_SIZE set and this function is never reached.)"""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is a pipe: it is drained while the ranks run (a line longer than the pipe buffer would otherwise block
    # rank 0 in write() while this process waits for it to exit)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    live = set(range(n_ranks))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.05)
    if failed is not None:
        for r in live:                       # exactly the children started above
            procs[r].send_signal(signal.SIGTERM)
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}\n")
        raise SystemExit(1)
    reader.join(timeout=30)
    out = b"".join(chunks).decode()
    sys.stdout.write(out)
    sys.stdout.flush()
    if not out.strip():
        raise SystemExit("bench.py: rank 0 printed no result line")


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)
    # the contract is ONE JSON line on stdout: libraries that write there (RCCL prints a version banner when its first
    # communicator is created) are pointed at stderr, the line goes to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device and args.backend == "nccl" and world > 1:
        raise SystemExit("--one-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    from monte_carlo_localization_amd import engine, maps, synth

    n = args.particles_per_gpu
    if args.map == "spielberg":
        m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
        true_pose = TRUE_POSE
    elif args.map == "fine025":
        m = maps.synthetic_fine025(maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz")))
        true_pose = TRUE_POSE
    else:
        m = maps.synthetic_levine()
        true_pose = (-34.0, -34.9, 0.0)          # corridor centre near the lower-left corner of the loop
    ang = synth.beam_angles(angle_step=args.beam_step)
    B = ang.size
    mode = engine.RESAMPLE_MULTINOMIAL if args.resample == "multinomial" else engine.RESAMPLE_SYSTEMATIC

    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    e = engine.Engine(max_particles=n, device=local_rank, seed=42, resample_mode=mode)
    e.set_map(m.data, m.resolution, m.origin_x, m.origin_y)
    e.set_beam_angles(ang)
    planned_kernel, planned_reason = e.planned_ray_kernel(n)      # known before the first update (mcl_get_planned_ray_kernel)
    # noise-free scan from the true pose: the committed fixture (tests assert the engine regenerates it
    # bit for bit); keeps the profiled run free of a stray 1-particle k_rays launch
    if args.map == "spielberg":
        scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)[::args.beam_step].copy()
    else:
        scan = synth.scan_from_pose(e, m, ang, true_pose)
    rng = np.random.default_rng(42 + rank)
    p = synth.tracking_cloud(rng, n, true_pose) if args.regime == "tracking" else synth.global_cloud(rng, m, n)
    sample_first = np.ascontiguousarray(p[:, :4000])           # the first update traces the spread cloud
    w0 = np.full(n, 1.0 / (n * world))
    e.set_particles(p, w0)
    # one throw-away update sizes every lazily allocated buffer (work lists, partial sums, tables), then the particle set is
    # put back: first_update_ms below is the cost of an update on the spread cloud, not of hipMalloc
    e.update(ACTION, scan)
    e.set_particles(p, w0)
    n_updates = 1                             # updates this engine has run: the Philox counter of the next one (E7)

    if use_dist:
        from monte_carlo_localization_amd.dist import ShardedFilter
        dev = torch.device("cuda", local_rank)
        sf = ShardedFilter(e, n, dev, overlap=not args.no_overlap)
        sf.update(ACTION, scan)               # the same for the exchange: RCCL builds its communicators on first use
        n_updates += 1
        e.set_particles(p, w0)
        sf.reset()

        def step():
            sf.update(ACTION, scan)
    else:
        def step():
            e.update(ACTION, scan)
    del p, w0

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: the FIRST update works on the spread cloud the workload string names; resampling then collapses the set to
    # motion-noise width around a few parents, which is the state every timed update sees.  Both are reported.
    first_ms, first_ray_ms = None, None
    for k in range(args.warmup):
        fence()
        t0 = time.perf_counter()
        step()
        n_updates += 1
        fence()
        if k == 0:
            first_ms, first_ray_ms = (time.perf_counter() - t0) * 1e3, e.ray_kernel_ms()
    ray_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ray_ms.append(e.ray_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    n_updates += args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pose = sf.expected_pose() if use_dist else e.expected_pose()
    counters = e.counters()
    exit_code = 0
    kernel_name = e.ray_kernel_name()
    kernel_variant = e.ray_kernel_variant() if kernel_name == "k_rays_sweep" else None      # (of the last TIMED update)

    # ---- after the timed region: the state the last timed update left, then ONE more untimed update (probe counter on) whose
    #      resample indices the oracle checks.  A sharded run gathers every rank's log-weights and children's parents on rank 0
    #      (all ranks take part in the two gathers), so that a line with N > 1 proves its indices like the single-GPU line does.
    n_total = n * world
    pt_last = e.get_particles() if rank == 0 else None
    lw_last = e.log_weights()
    lw_all, idx_all, probes_live = (lw_last if world == 1 else None), None, None
    if not args.no_parity_check:
        def gather_all(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            if args.backend == "nccl":
                t = t.to(torch.device("cuda", local_rank))
            out = torch.empty(t.numel() * world, dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t)
            return out.cpu().numpy() if rank == 0 else None
        if world > 1:
            lw_all = gather_all(lw_last)
        e.set_debug_count_probes(1)
        step()
        e.set_debug_count_probes(0)
        probes_live = e.counters()["probes"] / float(n * B)
        idx_mine = np.ascontiguousarray(e.resample_indices(), np.int32)
        idx_all = gather_all(idx_mine) if world > 1 else idx_mine

    if rank == 0:
        ms = elapsed * 1e3 / args.steps
        value = n * world * B / (elapsed / args.steps)
        base, sbar_first, sbar_timed = None, None, None
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(m, ang, scan, true_pose, big=args.cpu_baseline_256k)
        parity, sbar_err = None, None
        checker_failed = False
        orc = None
        try:
            # the oracle is test infrastructure: a box without it (no gcc, no oracle/) still gets its line, marked unchecked
            from oracle import oracle as orc
            orc.lib()
        except (ImportError, OSError, subprocess.CalledProcessError) as ex:
            orc, sbar_err = None, repr(ex)
        if orc is not None:
            try:
                # The oracle as the CHECKER of the run that was just timed (never inside the timed region): the log-weights the
                # last timed update left for 4000 sampled particles of rank 0 against orc_eng_log_weights on those particles (which
                # also gives S-bar, the reference's samples per ray, cpp:622-647), then the resample indices of the untimed update
                # above -- all children of all ranks -- against the oracle's exact-CDF draw from the weights the timed update left.
                # An oracle error or a mismatch from here on fails the run.
                om = orc.OracleMap(m.data, m.resolution, m.origin_x, m.origin_y)
                L = orc.eng_log_table(orc.sensor_table(om.max_range_px))
                oi = orc.obs_index(scan, om)
                pick = np.random.default_rng(7).choice(n, size=min(4000, n), replace=False)
                logw_o, _, probes_o = orc.eng_log_weights(om, np.ascontiguousarray(pt_last[:, pick]), ang, oi, L)
                sbar_timed = probes_o / float(pick.size * B)
                _, _, probes_f = orc.eng_log_weights(om, sample_first, ang, oi, L)
                sbar_first = probes_f / float(sample_first.shape[1] * B)
                parity = {"n": int(pick.size), "logw_mismatches": int(np.count_nonzero(lw_last[pick] != logw_o))}
                if idx_all is not None:
                    _, q_prev, _ = orc.eng_weights_from_log(lw_all)          # fixed-point weights of the WHOLE set (global maximum)
                    if mode == engine.RESAMPLE_MULTINOMIAL:
                        want = orc.eng_resample_indices(q_prev, 0, k53=orc.eng_philox_k53(42, n_updates, 0, n_total))
                    else:
                        want = orc.eng_resample_indices(q_prev, 1, k0=orc.eng_philox_k0(42, n_updates))
                    parity.update({"idx_n": int(n_total), "idx_mismatches": int(np.count_nonzero(idx_all != want)),
                                   "idx_scope": "all children of all ranks (global parent indices)" if world > 1 else "all children"})
                    del q_prev, want
            except Exception as ex:                  # noqa: BLE001 -- the checker failed: the line still goes out, marked, and the run fails
                parity = {"error": repr(ex)}
                checker_failed = True
        del pt_last, lw_all, idx_all
        k_ms = float(np.mean(ray_ms))
        roof = roofline_block(kernel_name, k_ms, n, B, sbar_timed if sbar_timed is not None else 43.4,
                              profiled_workload=(args.map == "spielberg" and args.regime == "tracking"), trips_live=probes_live,
                              variant=kernel_variant)
        # which kernel class AUTO gave this map / scan / size, and why, when it is not the fast windowed kernel
        roof["kernel_class_planned"] = planned_kernel
        if kernel_name != "k_rays_sweep":
            roof["kernel_class_reason"] = planned_reason
        roof["algorithmic"]["s_bar_first_update"] = sbar_first
        # what the kernel itself examines: loop trips per ray counted by an extra, untimed update (debug_count_probes)
        roof["probe_trips_per_ray_live"] = probes_live
        line = {
            "metric": "MCL updates/sec (particle*beam/s)",
            "value": value, "unit": "particle*beam/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n} particles/GPU x {B} beams, "
                                   + ("Spielberg_map (2000x2000 @ 0.05796), " if args.map == "spielberg"
                                      else "SYNTHETIC Spielberg_map grid split 2 x 2 per cell and declared 0.025 m (4000x4000, MAX_RANGE_PX 479), " if args.map == "fine025"
                                      else "SYNTHETIC levine stand-in (2049x2049 @ 0.05), ")
                                   + (f"initial cloud N({true_pose},(0.5 m,0.5 m,0.4 rad)) (tracking regime), " if args.regime == "tracking"
                                      else "initial cloud uniform over the free cells (global regime), ")
                                   + f"action {ACTION}, "
                                   f"stock sensor/motion params, {args.resample} resampling, Philox seed 42; the timed updates "
                                   f"follow {args.warmup} warm-up updates (the particle set has resampled to motion-noise width)",
                       "particles_total": n * world, "beams": B,
                       "parallelism": (f"particle-sharded x{world}" + (" (REHEARSAL: all ranks on one GPU over gloo)" if args.one_device else ""))
                                      if world > 1 else "single GPU"},
            "first_update_ms": first_ms, "first_update_ray_kernel_ms": first_ray_ms,
            "steady_state_ms": ms,
            "roofline": roof,
            "cpu_baseline": base,
            "pose": [float(v) for v in pose],
            "counters_last_update": counters,
            "parity_check": parity if parity is not None else {"skipped": sbar_err},
        }
        if use_dist:
            line["exchange_bytes_per_update_per_gpu"] = sf.exchange_bytes
            # stream synchronisations of one steady-state update on this rank (dist.py: 1 = device-ordered, the small exchanges
            # stay in device memory; MCL_DIST_SYNC=1: a wait per exchanged value)
            line["host_waits_per_update"] = sf.host_waits
            # who runs the collectives of an update: the engine itself (mcl_comm_*: RCCL on its own stream, one native call per
            # update: MCL_DIST_NATIVE=1) or dist.py through torch.distributed (the default)
            line["collectives"] = "engine (RCCL on the engine's stream)" if sf.native else f"torch.distributed ({args.backend})"
            # the failure protocol's error word of the last timed update (ranks that failed: 0 = a good update on every rank; a
            # non-zero word raises ShardedUpdateError on every rank, so a line that exists says 0) and the bound on a host wait
            line["failure_protocol"] = {"failed_ranks_last_update": int(sf.last_failed_ranks),
                                        "bounded_wait": "MCL_COMM_TIMEOUT_MS (30 s)" if sf.native else "process-group timeout"}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        if checker_failed:
            sys.stderr.write(f"bench.py: THE PARITY CHECKER FAILED {parity}\n")
            exit_code = 4
        elif parity and (parity["logw_mismatches"] or parity.get("idx_mismatches", 0)):
            sys.stderr.write(f"bench.py: PARITY CHECK FAILED {parity}\n")
            exit_code = 3
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    e.close()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
