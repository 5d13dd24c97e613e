#!/usr/bin/env python
"""Developer tool: package power and shader clock (rocm-smi, 4 Hz) while sp_pairs_cost runs back to back in one mode.
    python tools/power_by_mode.py --modes 1,11,13,0 [--seconds 6]
modes: 1 = GN pass, 0 = gradient pass, 11 / 10 = the same without the target gathers, 13 / 12 = loads + geometry only."""
import argparse, os, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="1,14,11,13")
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--pairs", type=int, default=384)
a = ap.parse_args()
args = argparse.Namespace(pairs=a.pairs, distinct=4, segments=64, tile_points=8192, span_points=None)
batch, _ = bench.build_batch(args, 0, torch.device("cuda:0"))


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = [l for l in txt.splitlines() if "sclk" in l.lower()]
            pw = [l for l in txt.splitlines() if "package power" in l.lower() or "socket power" in l.lower()]
            out.append((sclk[0].split("(")[-1].split("Mhz")[0] if sclk else "?", pw[0].split(":")[-1].strip() if pw else "?"))
        except Exception as e:
            out.append(("err", str(e)))
        time.sleep(0.25)


for mode in [int(m) for m in a.modes.split(",")]:
    for _ in range(50):
        batch.cost_pass(0, mode)
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(200):
            batch.cost_pass(0, mode)
        torch.cuda.synchronize(); n += 200
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    good = [(float(s), float(p)) for s, p in out[2:] if s not in ("?", "err") and p not in ("?",)]
    print(f"mode {mode}: {1e6 * dt / n:.1f} us/launch; sclk MHz median {np.median([g[0] for g in good]):.0f} (min {min(g[0] for g in good):.0f}, max {max(g[0] for g in good):.0f}); "
          f"power W median {np.median([g[1] for g in good]):.0f} (max {max(g[1] for g in good):.0f}); {len(good)} samples", flush=True)
