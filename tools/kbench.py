#!/usr/bin/env python
"""Kernel micro-bench (developer tool): times sp_pairs_cost (both modes) on a streaming batch with HIP events.

    python tools/kbench.py [--pairs 64] [--tile-points 2048] [--reps 20]
"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=2)
    ap.add_argument("--tile-points", type=int, default=2048)
    ap.add_argument("--segments", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--granule", type=int, default=256)
    ap.add_argument("--no-depth-table", action="store_true", help="log-depth tables (exp per point) instead of depth tables")
    ap.add_argument("--ab-depth-table", action="store_true", help="time mode 1 / mode 0 on a depth-table and a log-depth-table batch, interleaved")
    ap.add_argument("--shape", default="grid")
    ap.add_argument("--coverage", type=float, default=1.2)
    ap.add_argument("--ab-granule", action="store_true", help="time mode 1 / mode 0 on a 256-granule and a 64-granule (wave spans) batch, interleaved")
    ap.add_argument("--levels", default="0")
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--tile-order", default="", help="comma list of t: re-time with the points of every chunk re-ordered into t x t pixel "
                                                     "blocks (developer experiment on the table order; 0 = restore row-major)")
    a = ap.parse_args()
    if any(int(x) >= 2 for x in a.modes.split(",")):
        a.no_depth_table = True          # (mode 2 and the developer modes read log-depth tables)
    dev = torch.device("cuda:0")
    batch, _ = bench.build_batch(a, 0, dev)
    for _ in range(3):
        batch.gn_step(0)
    if a.ab_granule or a.ab_depth_table:
        import copy
        a2 = copy.copy(a)
        if a.ab_granule:
            a2.granule = 64 if a.granule == 256 else 256
        else:
            a2.no_depth_table = not a.no_depth_table
        other, _ = bench.build_batch(a2, 0, dev)
        for rnd in range(4):
            for bt in (batch, other):
                for mode in (1, 0):
                    for _ in range(3):
                        bt.cost_pass(0, mode)
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
                    torch.cuda.synchronize()
                    for e0, e1 in ev:
                        e0.record(); bt.cost_pass(0, mode); e1.record()
                    torch.cuda.synchronize()
                    ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
                    by = bt.algorithmic_bytes(0)
                    print(f"level 0 granule {bt.granule:3d} {'depth tables' if bt.depth_table else 'log-depth tables'} mode {mode} pairs {bt.M} spans {bt.n_spans}: median {np.median(ms)*1e3:.1f} us  min {ms.min()*1e3:.1f} us -> "
                          f"{by/np.median(ms)/1e6/8000*100:.1f}% of 8 TB/s  (padding {sum(bt.Ppads)/sum(bt.Ps):.4f})", flush=True)
        return
    orders = [None] + [int(t) for t in a.tile_order.split(",") if t]
    original = (batch.pix.clone(), {l: v.clone() for l, v in batch.src4.items()})
    for order in orders:
      if order is not None:
        batch.pix.copy_(original[0])
        for l in batch.src4:
            batch.src4[l].copy_(original[1][l])
        if order > 0:
            ch = batch.chunks.long()
            run = torch.repeat_interleave(torch.arange(ch.shape[0], device=dev), ch[:, 3])       # chunk of every table position
            w = batch.pix.long() & 0xffffffff
            col, row, real = w & 0xffff, (w >> 16) & 0x7fff, (w != 0)
            key = (((row // order) * 8192 + col // order) * order + row % order) * order + col % order
            key = torch.where(real, key, torch.full_like(key, (1 << 38) - 1)) + (run << 38)
            perm = torch.argsort(key)
            batch.pix.copy_(batch.pix[perm])
            for l in batch.src4:
                batch.src4[l].view(-1, 4).copy_(batch.src4[l].view(-1, 4)[perm])
            del key, perm, w, col, row, real, run
        print(f"--- table order: {order} x {order} pixel blocks inside every chunk" if order else "--- table order: row-major")
      for level in [int(x) for x in a.levels.split(",")]:
        for mode in [int(x) for x in a.modes.split(",")]:
              if mode == 16:
                  # developer mode 16 (sp_cost.hip ABL 5): no pix stream -- the pixel word rides in the low mantissa bits of the
                  # point's own colours (7 + 7 + 6 bits; colours change by < 2^-16 relative, irrelevant for a timing experiment)
                  assert bench.W <= 1024 and bench.H <= 512
                  keep = batch.src4[level].clone()
                  w = batch.pix.view(torch.int32).long() & 0xffffffff
                  v = (w & 0x3ff) | (((w >> 16) & 0x1ff) << 10) | ((w >> 31) << 19)
                  q = batch.src4[level].view(-1, 4).view(torch.int32)
                  q[:, 0] = ((q[:, 0].long() & ~0x7f) | (v & 0x7f)).int()
                  q[:, 1] = ((q[:, 1].long() & ~0x7f) | ((v >> 7) & 0x7f)).int()
                  q[:, 2] = ((q[:, 2].long() & ~0x3f) | ((v >> 14) & 0x3f)).int()
                  del w, v
              for _ in range(3):
                  batch.cost_pass(level, mode)
              ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
              torch.cuda.synchronize()
              for e0, e1 in ev:
                  e0.record(); batch.cost_pass(level, mode); e1.record()
              torch.cuda.synchronize()
              ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
              by = batch.algorithmic_bytes(level)
              if mode == 16:
                  batch.src4[level].copy_(keep)
                  del keep
              print(f"level {level} mode {mode} pairs {batch.M} spans {batch.n_spans}: median {np.median(ms)*1e3:.1f} us  min {ms.min()*1e3:.1f} us  "
                    f"alg {by/1e6:.1f} MB -> {by/np.median(ms)/1e6:.0f} GB/s ({by/np.median(ms)/1e6/8000*100:.1f}% of 8 TB/s)  "
                    f"{sum(batch.Ps)/np.median(ms)/1e6:.2f} Gpt/s")


if __name__ == "__main__":
    main()
