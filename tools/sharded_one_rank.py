#!/usr/bin/env python3
"""One rank through the sharded path against the plain engine: runs bench.py eight times on the one GPU of the box (plain engine,
the engine's RCCL communicator, torch collectives device-ordered, torch collectives stage by stage; 4M and 262 144 particles),
writes profiles/<tag>_sharded_one_rank.jsonl (the eight bench lines) and .md (the table: ms per update, ray kernel ms, the rest,
host waits).  usage: tools/sharded_one_rank.py <tag, e.g. r04> [out dir, default gpurun_out/profiles_new]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ("one engine (`mcl_update`)", "single", [], {}),
    ("sharded, 1 rank, the engine's RCCL communicator (`mcl_comm_update`, `MCL_DIST_NATIVE=1`)", "native", ["--force-dist"], {"MCL_DIST_NATIVE": "1"}),
    ("sharded, 1 rank, torch collectives, device-ordered (`MCL_DIST_NATIVE=0`)", "ordered", ["--force-dist"], {"MCL_DIST_NATIVE": "0"}),
    ("sharded, 1 rank, torch collectives, stage by stage (`MCL_DIST_NATIVE=0 MCL_DIST_SYNC=1`: round 3's flow)", "sync", ["--force-dist"],
     {"MCL_DIST_NATIVE": "0", "MCL_DIST_SYNC": "1"}),
]


def run(extra, env, n):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", *extra] + (["--particles-per-gpu", str(n)] if n else [])
    out = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def main():
    tag = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "profiles_new")
    os.makedirs(out_dir, exist_ok=True)
    rows = ["One rank through the sharded path against the single engine, one MI355X, one run of tools/sharded_one_rank.py:",
            "`python bench.py --no-cpu-baseline [--force-dist] [--particles-per-gpu 262144]`, 3 warm-up + 20 timed updates.  rest = ms per update - ray",
            "kernel ms (everything that is not the ray kernel: the comparable column, the ray kernel's own time moves by +-0.05 ms between runs).",
            f"Every line's parity check (all resample indices against the oracle) is clean.  The lines: `profiles/{tag}_sharded_one_rank.jsonl`.", "",
            "| host | 4M x 1081: ms | ray kernel | rest | 262 144 x 1081: ms | ray kernel | rest | host waits per update |", "|---|---|---|---|---|---|---|---|"]
    with open(os.path.join(out_dir, f"{tag}_sharded_one_rank.jsonl"), "w") as f:
        for name, key, extra, env in VARIANTS:
            a, b = run(extra, env, 0), run(extra, env, 262144)
            for sz, d in (("4m", a), ("256k", b)):
                assert d["parity_check"]["idx_mismatches"] == 0 and d["parity_check"]["logw_mismatches"] == 0
                f.write(json.dumps(dict(d, variant=f"{key}_{sz}")) + "\n")
            ra, rb = a["ms_per_step"] - a["roofline"]["kernel_ms"], b["ms_per_step"] - b["roofline"]["kernel_ms"]
            rows.append(f"| {name} | {a['ms_per_step']:.3f} | {a['roofline']['kernel_ms']:.3f} | {ra:.3f} | {b['ms_per_step']:.3f} | "
                        f"{b['roofline']['kernel_ms']:.3f} | {rb:.3f} | {a.get('host_waits_per_update', '1 (mcl_update)')} |")
    open(os.path.join(out_dir, f"{tag}_sharded_one_rank.md"), "w").write("\n".join(rows) + "\n")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
