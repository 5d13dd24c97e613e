#!/usr/bin/env python
"""The verdict of the frame-pair schedule over MANY of the reference's own starts: does any pair that ends away
from its ground truth leave the run UNFLAGGED, what does each status bit catch, what does the second attempt bring home.

    python tools/verdict_sweep.py [--starts 9216] [--batch 1536] [--slots 384] [--variants shipped,no_retry,...] [--npz out.npz] [--shape grid|blobs]

Starts are bench.py's reference-start leg continued: scenes 5000..5007 (640x480x64, multi-octave texture), start m = scene m % 8 with
pose T_gt Exp(0.05 xi), depth seeds log(2 + 2 u) drawn from default_rng(77) in bench's order (m < 8: the scene's own), so pair m here
IS bench pair m (105, 1380, 1482, ... of DESIGN.md section 6).  Per variant: pairs/s, iterations per pair, misses against the ground
truth (golden g19's criterion: 2e-3 rad / 2e-3 t / 2e-2 depth), how many of those the verdict flags (SILENT = missed and not flagged --
must be 0), false alarms, second attempts made and what they rescued.  ``--npz`` keeps the per-pair arrays of every variant.
"""
from __future__ import annotations

import argparse
import copy
import os
import sys
import time
from multiprocessing import Pool

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from super_primitive_amd import _lib, synth  # noqa: E402
from super_primitive_amd.image.keyframe import KeyFrame  # noqa: E402
from super_primitive_amd.optim.pair_batch import (REFERENCE_START_LEVELS, REFERENCE_START_POINT_STRIDE, REFERENCE_START_SCHEDULE,  # noqa: E402
                                                  VERDICT_DEFAULTS, PairBatch)

H, W, N, G = 480, 640, 64, 8
BASE = {k: v for k, v in REFERENCE_START_SCHEDULE.items() if k != "check_every"}
NO_CAP = VERDICT_DEFAULTS["retry_on"] & ~_lib.SP_STATUS_LAST_CAP
ct, ie = BASE["conv_tol"], 1e-3
jt = lambda level, stride, damp=0.0, cap=25: dict(level=level, stride=stride, max_iters=cap, irls_eps=ie, conv_tol=ct, depth_damp=damp)
po = lambda level, stride, cap, eps=ie: dict(level=level, stride=stride, max_iters=cap, irls_eps=eps, conv_tol=ct, pose_only=True)
pol = dict(level=0, stride=1, max_iters=15, irls_eps=1e-5, conv_tol=1e-4)
def damped(damp, cap, tol=ct, second_L2=True, eps=ie):
    return [po(2, 4, 15, 1e-2), dict(level=2, stride=4, max_iters=cap, irls_eps=eps, conv_tol=tol, depth_damp=damp)] + ([jt(2, 4)] if second_L2 else []) + [jt(1, 2), jt(0, 2), pol]
adam = lambda level, stride, n=500, eps=1e-5: dict(level=level, stride=stride, max_iters=n, irls_eps=eps, conv_tol=0.0, adam=True)
VARIANTS = {
    "shipped": dict(BASE),
    "pred": dict(BASE, predicted_exit=True),                                          # SP_PHASE_PREDICTED_EXIT in every phase
    "pred_tight": dict(BASE, predicted_exit=True, conv_tol=1e-3, polish_tol=5e-5),
    "two_attempts": dict(BASE, retry2_phases=None),                                   # round 5's: no third attempt
    "adam_all_points": dict(BASE, retry2_phases=[adam(2, 1), adam(1, 1), adam(0, 1)]),  # the reference's own point set at every level
    "adam_L2_L1": dict(BASE, retry2_phases=[adam(2, 4), adam(1, 2)]),
    "adam_300": dict(BASE, retry2_phases=[adam(2, 4, 300), adam(1, 2, 300), adam(0, 2, 300)]),
    "adam_eps1e-3": dict(BASE, retry2_phases=[adam(2, 4, 500, 1e-3), adam(1, 2, 500, 1e-3), adam(0, 2, 500, 1e-3)]),
    "no_retry": dict(BASE, retry_phases=None, retry2_phases=None),
    "undamped": dict(BASE, phases=None, depth_damp=None, coarse_damped=None),
    "d8c12": dict(BASE, coarse_damped=(8.0, 12)),
    "d12c12": dict(BASE, coarse_damped=(12.0, 12)),
    "d20c12": dict(BASE, coarse_damped=(20.0, 12)),
    "d24c12": dict(BASE, coarse_damped=(24.0, 12)),
    "d16c12": dict(BASE, coarse_damped=(16.0, 12)),
    "d16c16": dict(BASE, coarse_damped=(16.0, 16)),
    "d31c12": dict(BASE, coarse_damped=(31.0, 12)),
    "d31c16": dict(BASE, coarse_damped=(31.0, 16)),
    "d16c12_retry_d31c16": dict(BASE, coarse_damped=(16.0, 12), retry_phases=[po(2, 4, 30), jt(2, 4, 31.0, 16)], retry_join=2),
    "d16c12_retry_d8c16": dict(BASE, coarse_damped=(16.0, 12), retry_phases=[po(2, 4, 30), jt(2, 4, 8.0, 16)], retry_join=2),
    "d16c16_retry_d31c16": dict(BASE, coarse_damped=(16.0, 16), retry_phases=[po(2, 4, 30), jt(2, 4, 31.0, 16)], retry_join=2),
}


def _render(a):
    shape_kw = dict(overlap=4) if a[1] == "grid" else dict(shape=a[1], blob_coverage=1.2)
    return synth.make_pair(H, W, a[2], seed=a[0], init_sigma=0.05, texture="octaves", init_mode="reference", **shape_kw)


observable_segments = synth.observable_segments


def errors(P, K, poses_gt, klds_gt, observable=None):
    """(n, 3): rotation [rad], translation [max abs, scale gauge removed], depth [max relative over the segments the target frame sees]
    against the ground truth."""
    out = np.zeros((len(P), 3))
    for m in range(len(P)):
        gt_T, gt_k = poses_gt[m], klds_gt[m]
        if observable is not None:
            gt_k, Km = gt_k[observable[m]], K[m][observable[m]]
            K = list(K); K[m] = Km
        ls = float(np.mean(gt_k - K[m]))
        Rm = P[m][:3, :3].T @ gt_T[:3, :3]
        out[m] = (float(np.arctan2(0.5 * np.linalg.norm([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]]), 0.5 * (np.trace(Rm) - 1))),
                  float(np.abs(P[m][:3, 3] * np.exp(ls) - gt_T[:3, 3]).max()), float(np.abs(np.expm1(K[m] + ls - gt_k)).max()))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--starts", type=int, default=9216)
    ap.add_argument("--batch", type=int, default=1536)
    ap.add_argument("--slots", type=int, default=384)
    ap.add_argument("--variants", default="shipped,no_retry")
    ap.add_argument("--shape", default="grid", choices=["grid", "blobs", "sam"])
    ap.add_argument("--npz", default=None)
    ap.add_argument("--segments", type=int, default=N, help="segments per keyframe (bench.py --segments)")
    ap.add_argument("--alone", default="105,1380,1482", help="pairs also run as batches of ONE (order independence of the verdict and the retry)")
    ap.add_argument("--seed0", type=int, default=5000, help="first scene seed (5000 = bench.py's scenes, what the schedule and the verdict's thresholds were "
                                                            "tuned on; any other value = HELD-OUT scenes)")
    ap.add_argument("--streams", type=int, default=1, help="run_scheduled(streams=...): groups of slots on their own HIP streams, one queue")
    ap.add_argument("--verdict", default="", help="overrides of VERDICT_DEFAULTS, e.g. seg_max_ratio=4,seg_mean_ratio=1.3,cost_outlier=0")
    ap.add_argument("--only-batches", default="", help="comma list of batch indices to run (the others are skipped): the known hard starts' batches")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    t0 = time.time()
    with Pool(min(G, 8)) as pool:
        scenes = pool.map(_render, [(args.seed0 + s, args.shape, args.segments) for s in range(G)])
    print(f"rendered {G} scenes ({args.shape}, seeds {args.seed0} .. {args.seed0 + G - 1}{'' if args.seed0 == 5000 else ': HELD OUT -- not the scenes anything was tuned on'}) in {time.time() - t0:.1f} s", flush=True)
    rng = np.random.default_rng(77)
    n_batches = -(-args.starts // args.batch)
    total = n_batches * args.batch
    poses, klds = [], []
    for r in range(total // G):
        for p in scenes:
            if r == 0:
                poses.append(p.pose_init); klds.append(p.kld_init)
            else:
                poses.append((p.pose_gt.astype(np.float64) @ synth.se3_exp_np(0.05 * rng.standard_normal(6))).astype(np.float32))
                klds.append(np.log(2.0 + 2.0 * rng.uniform(size=p.N)).astype(np.float32))
    seen = [observable_segments(p) for p in scenes]
    print("segments the target frame does not see (left out of the depth error): " + ", ".join(f"scene {args.seed0 + i}: {int((~o).sum())} of {len(o)}" for i, o in enumerate(seen)), flush=True)
    poses_gt = [scenes[m % G].pose_gt.astype(np.float64) for m in range(total)]
    klds_gt = [scenes[m % G].kld_gt.astype(np.float64) for m in range(total)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    src = [KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions)) for p in scenes]
    trg, Ks = [t(p.trg_image) for p in scenes], [t(p.K) for p in scenes]
    names = [v for v in args.variants.split(",") if v]
    verdict = {k: float(x) for k, x in (kv.split("=") for kv in args.verdict.split(",") if kv)} or None
    only = {int(b) for b in args.only_batches.split(",") if b}
    keep = {v: dict(err=[], status=[], diag=[], attempts=[], iters=[], kld=[], secs=0.0, rounds=0) for v in names}
    ran = []
    for b in range(n_batches):
        if only and b not in only:
            continue
        ran.append(b)
        lo = b * args.batch
        batch = PairBatch(src, trg, Ks, torch.from_numpy(np.stack(poses[lo: lo + args.batch])), [t(k) for k in klds[lo: lo + args.batch]],
                          levels=REFERENCE_START_LEVELS, replicate=args.batch // G, point_stride=REFERENCE_START_POINT_STRIDE, granule=64)
        for v in names:
            kw = dict(VARIANTS[v])
            for rep in range(2 if b == ran[0] else 1):             # (first use untimed)
                batch.restore_initial()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rounds = batch.run_scheduled(slots=args.slots, verdict=verdict, streams=args.streams, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
            P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
            k = keep[v]
            k["err"].append(errors(P, K, poses_gt[lo: lo + args.batch], klds_gt[lo: lo + args.batch], [seen[(lo + i) % G] for i in range(args.batch)]))
            k["status"].append(batch.status.cpu().numpy().copy()); k["diag"].append(batch.diag.cpu().numpy().copy())
            k["attempts"].append(batch.attempts.cpu().numpy().copy())
            k["kld"].extend(K)
            k["iters"].append((batch.lm_state[:, 2] + batch.lm_state[:, 3]).cpu().numpy())
            k["secs"] += dt; k["rounds"] += rounds
        if b == 0 and args.alone:
            # order independence: the named pairs alone (other span partition, other summation order) and all resident
            ids = [int(a) for a in args.alone.split(",") if int(a) < args.batch]
            kw = dict(VARIANTS[names[0]])
            batch.restore_initial()
            batch.run_scheduled(verdict=verdict, **kw)
            torch.cuda.synchronize()
            P, K = batch.poses().double().cpu().numpy(), [k.double().cpu().numpy() for k in batch.klds()]
            e = errors([P[m] for m in ids], [K[m] for m in ids], [poses_gt[m] for m in ids], [klds_gt[m] for m in ids])
            st = batch.status.cpu().numpy()
            print(f"all {args.batch} resident ({names[0]}): " + "; ".join(f"pair {m}: {e[i]} status {st[m]:#x}" for i, m in enumerate(ids)), flush=True)
            for m in ids:
                one = PairBatch([src[m % G]], [trg[m % G]], [Ks[m % G]], torch.from_numpy(poses[m][None]), [t(klds[m])], levels=REFERENCE_START_LEVELS,
                                point_stride=REFERENCE_START_POINT_STRIDE, granule=64)
                one.run_scheduled(verdict=verdict, **kw)
                torch.cuda.synchronize()
                e1 = errors([one.poses()[0].double().cpu().numpy()], [one.klds()[0].double().cpu().numpy()], [poses_gt[m]], [klds_gt[m]])[0]
                print(f"pair {m} ALONE ({names[0]}): {e1} status {int(one.status[0]):#x} attempts {int(one.attempts[0])} diag {one.diag[0].cpu().numpy()}", flush=True)
                del one
        del batch
        torch.cuda.empty_cache()
        print(f"batch {b + 1}/{n_batches} done ({time.time() - t0:.0f} s; last timed run {1e3 * dt:.1f} ms, {rounds} rounds)", flush=True)
    out = {}
    F = _lib.SP_STATUS_FAILED
    index = np.concatenate([np.arange(b * args.batch, (b + 1) * args.batch) for b in ran])      # global pair index of every row
    total = len(index)
    for v in names:
        k = keep[v]
        err, st, dg, at, its = (np.concatenate(k[x]) for x in ("err", "status", "diag", "attempts", "iters"))
        miss = ~((err[:, 0] <= 2e-3) & (err[:, 1] <= 2e-3) & (err[:, 2] <= 2e-2))
        flagged = (st & F) != 0
        silent = np.nonzero(miss & ~flagged)[0]
        bits = {name: int(((st & getattr(_lib, "SP_STATUS_" + name)) != 0).sum()) for name in ("NONFINITE", "LAST_CAP", "DEPTH_RANGE", "COST", "VALID", "SEGMENTS", "RETRIED", "ADAM", "UNFINISHED")}
        conv = ~miss
        print(f"\n== {v}: {total} starts, {total / k['secs']:.0f} pairs/s ({args.slots} slots), {its.mean():.1f} iterations per pair, {k['rounds']} rounds\n"
              f"   missed (vs ground truth) {int(miss.sum())}: {index[miss][:24].tolist()}\n"
              f"   flagged {int(flagged.sum())} (of the missed: {int((miss & flagged).sum())}; FALSE ALARMS {int((flagged & ~miss).sum())}: {index[flagged & ~miss][:16].tolist()})\n"
              f"   SILENT {len(silent)}: {index[silent][:24].tolist()}\n"
              f"   later attempts {int((at > 0).sum())}: rescued {int(((at > 0) & conv & ~flagged).sum())}, converged but still flagged {int(((at > 0) & conv & flagged).sum())}, "
              f"missed again {int(((at > 0) & miss).sum())}; THIRD attempts {int((at > 1).sum())}: {index[at > 1][:24].tolist()}, rescued {int(((at > 1) & conv & ~flagged).sum())}\n"
              f"   status bits {bits}\n"
              f"   worst error of the unflagged: {err[~flagged].max(axis=0) if (~flagged).any() else None}", flush=True)
        # the within-pair statistics (diag[6] = median, diag[7] = maximum over the pair's segments of the segment's mean |r|)
        judged = dg[:, 6] > 0
        if judged.any():
            r_max, r_mean = dg[:, 7] / np.maximum(dg[:, 6], 1e-30), dg[:, 0] / np.maximum(dg[:, 6], 1e-30)
            good = judged & conv & ~flagged
            q = lambda a: "p50 %.2f p99 %.2f p99.9 %.2f p99.99 %.2f max %.2f" % tuple(np.percentile(a, [50, 99, 99.9, 99.99, 100])) if len(a) else "--"
            score = (r_mean - 1.0) * r_max
            print(f"   segment costs, converged unflagged pairs ({int(good.sum())}): worst / median {q(r_max[good])}; cost / median {q(r_mean[good])}; "
                  f"(cost / median - 1) x (worst / median) {q(score[good])}\n"
                  f"   ... pairs that END away from the ground truth ({int((judged & miss).sum())}): worst / median {np.sort(r_max[judged & miss])[:16].round(2).tolist()}; "
                  f"cost / median {np.sort(r_mean[judged & miss])[:16].round(2).tolist()}; product {np.sort(score[judged & miss])[:16].round(2).tolist()}", flush=True)
        kl = k["kld"]
        for m in list(silent[:8]) + list(np.nonzero(flagged & ~miss)[0][:4]):
            sc = scenes[m % G]
            rel = np.abs(np.expm1(kl[m] + float(np.mean(sc.kld_gt - kl[m])) - sc.kld_gt))
            worst = np.argsort(rel)[::-1][:3]
            px = sc.keypoint_regions.reshape(sc.N, -1).sum(1)
            print(f"   {'silent' if m in silent else 'false alarm'} pair {index[m]}: err {err[m]} status {st[m]:#x} diag {dg[m]}; worst segments {worst.tolist()} "
                  f"depth errors {rel[worst]} pixels {px[worst].tolist()} (median segment {int(np.median(px))} px)")
        for m in np.nonzero(miss & flagged)[0][:8]:
            print(f"   flagged miss {index[m]}: err {err[m]} status {st[m]:#x} attempts {at[m]} diag {dg[m]}")
        out.update({f"{v}__err": err, f"{v}__status": st, f"{v}__diag": dg, f"{v}__attempts": at, f"{v}__iters": its})
    if args.npz:
        os.makedirs(os.path.dirname(os.path.abspath(args.npz)), exist_ok=True)
        np.savez_compressed(args.npz, **out)


if __name__ == "__main__":
    main()
