import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repository root
sys.path.insert(0, ROOT)
os.environ["MCL_DEBUG_WG"] = "/tmp/wg.bin"
from monte_carlo_localization_amd import engine, maps, synth
m = maps.load_npz(os.path.join(ROOT, "tests", "golden", "map_Spielberg_map.npz"))
ang = synth.beam_angles()
n = int(os.environ.get("WG_N", 4 << 20))
e = engine.Engine(max_particles=n, seed=42)
e.set_map(m.data, m.resolution, m.origin_x, m.origin_y); e.set_beam_angles(ang)
scan = np.load(os.path.join(ROOT, "tests", "golden", "scan_Spielberg_map_origin.npz"))["ranges"].astype(np.float32)
regime = sys.argv[1] if len(sys.argv) > 1 else "tracking"      # usage: wg_stamps.py [tracking|global]
p = synth.tracking_cloud(np.random.default_rng(42), n) if regime == "tracking" else synth.global_cloud(np.random.default_rng(42), m, n)
e.set_particles(p, np.full(n, 1.0 / n))
for k in range(6):
    e.update((0.05, 0.0, 0.01), scan)
    d = np.fromfile("/tmp/wg.bin", dtype=np.uint64).reshape(-1, 4)
    t0, t1, items, bar = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64), d[:, 2] & np.uint64(0xFFFFFFFF), (d[:, 2] >> np.uint64(32)).astype(np.int64) / 100.0
    T0 = t0.min()
    dur = (t1.max() - T0) / 100.0          # us (100 MHz)
    end = (t1 - T0) / 100.0
    start = (t0 - T0) / 100.0
    wait = d[:, 3].astype(np.int64) / 100.0; busy = (t1 - t0).sum() / 100.0
    print(f"update {k}: kernel span {dur:.0f} us, ray_ms {e.ray_kernel_ms():.3f}; WG start spread {start.max():.0f} us; WG end: min {end.min():.0f} p10 {np.percentile(end,10):.0f} p50 {np.percentile(end,50):.0f} p90 {np.percentile(end,90):.0f} max {end.max():.0f}; "
          f"mean idle at end {(dur-end).mean():.0f} us ({(dur-end).mean()/dur:.1%}); items/WG min {items.min()} max {items.max()} total {items.sum()}; window phase per WG mean {wait.mean():.0f} us ({wait.mean()/dur:.1%}), of which first barrier {bar.mean():.0f} us", flush=True)
