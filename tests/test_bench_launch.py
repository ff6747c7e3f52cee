"""bench.py --gpus N without a launcher: the parent starts the N ranks itself before touching the GPU, hands rank 0's line
through and fails when a rank fails.  Exercised here with the rank body replaced by a stub (no GPU in this container)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_stub(tmp_path, body, gpus):
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the rank body (everything main() does once WORLD_SIZE is set) is replaced; parse_args / launch_ranks stay bench.py's own
    head, _ = src.split("    # the contract is ONE JSON line on stdout", 1)
    stub = head + textwrap.indent(textwrap.dedent(body), "    ") + "\n\nif __name__ == \"__main__\":\n    main()\n"
    f = tmp_path / "bench_stub.py"
    f.write_text(stub)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, str(f), "--gpus", str(gpus), "--steps", "2"], capture_output=True, text=True, env=env, timeout=120)


def test_parent_spawns_ranks_and_forwards_rank0_line(tmp_path):
    r = run_stub(tmp_path, """
        import json
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert int(os.environ["LOCAL_RANK"]) == rank and world == args.gpus
        assert "torch" not in sys.modules
        print(json.dumps({"rank": rank, "n_gpus": world, "steps": args.steps}))
        """, 4)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                       # ranks 1..3 write to stderr, rank 0's line is the only stdout
    assert json.loads(lines[0]) == {"rank": 0, "n_gpus": 4, "steps": 2}


def test_parent_fails_when_a_rank_fails(tmp_path):
    r = run_stub(tmp_path, """
        import time
        if int(os.environ["RANK"]) == 1:
            raise SystemExit(7)
        time.sleep(30)
        print("{}")
        """, 2)
    assert r.returncode != 0
    assert "rank 1 exited with code 7" in r.stderr


def test_single_gpu_call_runs_in_process():
    """--gpus 1 (the driver's N = 1 form) never spawns: parse_args defaults, no WORLD_SIZE needed."""
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("bench")
    a = b.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.particles_per_gpu) == (1, 20, 3, 4 * 1024 * 1024)


import pytest


@pytest.mark.gpu
def test_bench_gpus_2_end_to_end_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` as the driver calls it for N = 1 -- no launcher -- with the rehearsal switches (both ranks on
    GPU 0, gloo instead of RCCL, which refuses two ranks on one device): the parent spawns the ranks, they run the sharded
    update (dense first, then list exchanges), rank 0's line comes back with the run's own parity check green."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--particles-per-gpu", "131072", "--beam-step", "4", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["particles_total"] == 262144 and d["config"]["beams"] == 271
    assert d["parity_check"]["logw_mismatches"] == 0 and d["parity_check"]["n"] == 4000
    # ... including the resample indices of ALL children of BOTH ranks (global parent indices gathered on rank 0) against the
    # oracle's exact-CDF draw from the whole set's weights: a line with N > 1 proves its indices like the one-GPU line does
    assert d["parity_check"]["idx_n"] == 262144 and d["parity_check"]["idx_mismatches"] == 0
    assert d["roofline"]["probe_trips_per_ray_live"] > 1.0 and 0.0 < d["roofline"]["useful"]["frac"] < 1.0
    assert d["exchange_bytes_per_update_per_gpu"]["kind"] == "lists"
    assert d["value"] > 0 and d["roofline"]["kernel"] == "k_rays_sweep"
