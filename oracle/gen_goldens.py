"""Generate tests/golden/*.npz by running the REAL reference on seeded inputs.

TEST INFRASTRUCTURE.  Runs only in the build container, where /root/reference
exists; the fixtures it writes (inputs + the reference's outputs, data only)
are committed and travel to the GPU box, the reference does not.

    python oracle/gen_goldens.py            # rewrites tests/golden/

Import recipe: torchvision / lietorch are not installed and
are only touched by functions the hot path never calls, so they are stubbed as
empty modules before importing the reference packages.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "lietorch"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    # frontend/segment/post_processer.py needs cupy only for asarray/asnumpy and cupyx.scipy.ndimage.label, which
    # mirrors scipy.ndimage.label: stub them with numpy / scipy so the REAL module runs on the CPU
    import scipy.ndimage
    cp = types.ModuleType("cupy"); cp.asarray = np.asarray; cp.asnumpy = np.asarray
    cpx = types.ModuleType("cupyx"); cps = types.ModuleType("cupyx.scipy"); cps.ndimage = scipy.ndimage; cpx.scipy = cps
    sys.modules.update({"cupy": cp, "cupyx": cpx, "cupyx.scipy": cps, "cupyx.scipy.ndimage": scipy.ndimage})
    sys.path.insert(0, REF)
    import core.dense_optim as rdo
    import core.dense_optim_batch as rdob
    import core.depth_render as rdr
    import image.keyframe as rkf
    import lie.lie_algebra as rla
    import odometery.depth_init as rdi
    import tool.point_utils as rpu
    import frontend.segment.post_processer as rpp
    import odometery.kf_criteria as rkc
    import image.gaussian_pyramid as rgp
    import core.normal_cost as rnc
    import tool.etc as retc
    return types.SimpleNamespace(do=rdo, dob=rdob, dr=rdr, kf=rkf, la=rla, di=rdi, pu=rpu, pp=rpp, kc=rkc, gp=rgp, nc=rnc, etc=retc)


sys.path.insert(0, ROOT)
from super_primitive_amd import synth  # noqa: E402
from oracle import photometric_oracle as orc  # noqa: E402  (only for se3_exp inside the restated driver loops)


def T(a, dtype=torch.float32):
    # always a private copy: optimisers below update tensors in place and must not touch the arrays that get saved
    return torch.from_numpy(np.array(a, copy=True)).to(dtype)


def ref_frames(ref, pair, level_images=None):
    src_img = T(pair.src_image) if level_images is None else level_images[0]
    trg_img = T(pair.trg_image) if level_images is None else level_images[1]
    src = ref.kf.KeyFrame(src_img, T(pair.K), T(pair.logdepth_perseg), T(pair.keypoints),
                          torch.from_numpy(pair.keypoint_regions))
    trg = ref.kf.KeyFrame(trg_img, T(pair.K))
    return src, trg


def npify(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            out[k] = v.detach().cpu().numpy()
        elif isinstance(v, (tuple, list, torch.Size)):
            out[k] = np.asarray(v)
        else:
            out[k] = np.asarray(v)
    return out


def pair_inputs(pair):
    return dict(in_src_image=pair.src_image, in_trg_image=pair.trg_image, in_K=pair.K,
                in_logdepth=pair.logdepth_perseg, in_keypoints=pair.keypoints,
                in_masks=np.packbits(pair.keypoint_regions, axis=-1), in_HWN=np.array([pair.H, pair.W, pair.N]))


def big_rotation_pose(pair, pitch):
    """Target camera pitched by ~80 degrees: part of the points end up behind it (z' < 0), part project
    outside the frame, the rest stay valid (696 / 2376 / 480 of 3072 for the committed case)."""
    T = np.eye(4, dtype=np.float32)
    c, s = np.cos(pitch), np.sin(pitch)
    T[:3, :3] = [[1, 0, 0], [0, c, -s], [0, s, c]]
    T[:3, 3] = [0.05, 2.6, 0.3]
    return T


# ----------------------------------------------------------------------------
def golden_cost(ref, name, pair, pose, affine=None, levels=None):
    """G1+G2: forward (residual + stats at collect_stats=2) and autograd grads of residual.abs().mean()."""
    cases = []
    src_full, trg_full = ref_frames(ref, pair)
    if levels is None:
        frames = [(src_full, trg_full)]
    else:
        sp = ref.kf.keyframe_pyramid(src_full, levels[0], levels[1])
        tp = ref.kf.keyframe_pyramid(trg_full, levels[0], levels[1])
        frames = list(zip(sp, tp))
    save = pair_inputs(pair)
    save["in_pose"] = pose
    save["in_kld"] = pair.kld_init
    save["n_levels"] = np.array(len(frames))
    if affine is not None:
        save["in_aff_src"], save["in_aff_trg"] = affine
    for li, (s, t) in enumerate(frames):
        kld = T(pair.kld_init).requires_grad_(True)
        P = T(pose).requires_grad_(True)
        aff = None
        if affine is not None:
            aff = (T(affine[0]).requires_grad_(True), T(affine[1]).requires_grad_(True))
        cfg = {"mode": "colour", "collect_stats": 2}
        out = ref.do.photomeric_cost(s, t, kld, P, cfg, affine_comp=aff)
        loss = out["residual"].abs().mean()
        loss.backward()
        rec = npify(out)
        rec["g_kld"] = kld.grad.numpy()
        rec["g_pose"] = P.grad.numpy()
        if aff is not None:
            rec["g_aff_src"] = aff[0].grad.numpy()
            rec["g_aff_trg"] = aff[1].grad.numpy()
        rec["lvl_src_image"] = s.image.numpy()
        rec["lvl_trg_image"] = t.image.numpy()
        rec["lvl_K_img"] = s.K_img.numpy()
        for k, v in rec.items():
            save[f"L{li}_{k}"] = v
        cases.append(rec)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)
    return cases


def golden_precomputed(ref, name, pair, pose, affine):
    """G3: unproject_kf dict + photomeric_cost_precomputed residual and grads (pose, affine)."""
    src, trg = ref_frames(ref, pair)
    with torch.no_grad():
        pre = ref.do.unproject_kf(src, T(pair.kld_init))
    P = T(pose).requires_grad_(True)
    a0 = T(affine[0]).requires_grad_(True)
    a1 = T(affine[1]).requires_grad_(True)
    out = ref.do.photomeric_cost_precomputed(pre, trg, P, {"mode": "colour", "collect_stats": 0}, affine_comp=(a0, a1))
    out["residual"].mean().backward()
    save = pair_inputs(pair)
    save.update(in_pose=pose, in_kld=pair.kld_init, in_aff_src=affine[0], in_aff_trg=affine[1])
    save.update({"pre_" + k: v for k, v in npify(pre).items()})
    save.update(residual=out["residual"].detach().numpy(), g_pose=P.grad.numpy(), g_aff_src=a0.grad.numpy(),
                g_aff_trg=a1.grad.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_batch(ref, name, pair, B=3, seed=5):
    """G4: photomeric_cost_batch with B targets, distinct Ks / poses / affines."""
    rng = np.random.default_rng(seed)
    src, _ = ref_frames(ref, pair)
    imgs, Ks, poses = [], [], []
    for b in range(B):
        other = synth.make_pair(pair.H, pair.W, pair.N, seed=pair.meta["seed"], motion_scale=0.6 + 0.5 * b)
        imgs.append(other.trg_image)
        Kb = pair.K.copy()
        Kb[0, 0] *= 1.0 + 0.03 * b
        Kb[1, 1] *= 1.0 - 0.02 * b
        Kb[0, 2] += 0.7 * b
        Kb[1, 2] -= 0.4 * b
        Ks.append(Kb)
        poses.append((synth.se3_exp_np(0.02 * rng.standard_normal(6)) @ other.pose_gt.astype(np.float64)).astype(np.float32))
    imgs, Ks, poses = np.stack(imgs), np.stack(Ks), np.stack(poses)
    aff_s = np.array([0.03, -0.01], np.float32)
    aff_t = (0.05 * rng.standard_normal((B, 2))).astype(np.float32)
    kld = T(pair.kld_init).requires_grad_(True)
    P = T(poses).requires_grad_(True)
    a0 = T(aff_s).requires_grad_(True)
    a1 = T(aff_t).requires_grad_(True)
    out = ref.dob.photomeric_cost_batch(src, T(imgs), T(Ks), kld, P, {"mode": "colour", "collect_stats": 1},
                                        affine_comp=(a0, a1))
    (out["residual"] * T(np.arange(1, B + 1, dtype=np.float32))).sum().backward()
    save = pair_inputs(pair)
    save.update(in_trg_images=imgs, in_trg_Ks=Ks, in_poses=poses, in_kld=pair.kld_init, in_aff_src=aff_s,
                in_aff_trg=aff_t)
    save.update(npify(out))
    save.update(g_kld=kld.grad.numpy(), g_pose=P.grad.numpy(), g_aff_src=a0.grad.numpy(), g_aff_trg=a1.grad.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_pyramid(ref, name):
    """G5: keyframe_pyramid image levels + K_img for even and odd sizes."""
    save = {}
    for tag, (H, W) in {"even": (48, 64), "odd": (45, 67)}.items():
        pair = synth.make_pair(H, W, 4, seed=11)
        src, _ = ref_frames(ref, pair)
        for (s, e) in ((0, 3), (1, 4), (0, 1)):
            pyr = ref.kf.keyframe_pyramid(src, s, e)
            save[f"{tag}_{s}_{e}_n"] = np.array(len(pyr))
            for i, k in enumerate(pyr):
                save[f"{tag}_{s}_{e}_img{i}"] = k.image.numpy()
                save[f"{tag}_{s}_{e}_Kimg{i}"] = k.K_img.numpy()
                save[f"{tag}_{s}_{e}_K{i}"] = k.K.numpy()
        save[f"{tag}_image"] = pair.src_image
        save[f"{tag}_Kin"] = pair.K
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_depth_render(ref, name):
    """G6: estimate_depth_kf_native.  (a) fronto-parallel constant-depth segments moved by exactly half a
    pixel: truncation is unambiguous and no two points collide -> exact comparison.  (b) a general pose:
    compared statistically (scatter order on collisions is undefined in the reference)."""
    H, W, N = 40, 56, 4
    pair = synth.make_pair(H, W, N, seed=3)
    L = np.zeros_like(pair.logdepth_perseg)
    kld = np.log(np.array([2.0, 2.5, 3.0, 3.5], np.float32))
    src = ref.kf.KeyFrame(T(pair.src_image), T(pair.K), T(L), T(pair.keypoints), torch.from_numpy(pair.keypoint_regions))
    # per-segment depth differs, so use a pure half-pixel *rotation-free* shift scaled for segment 0 only
    pose = np.eye(4, dtype=np.float32)
    img_id = ref.dr.estimate_depth_kf_native(src, T(kld))
    src2, _ = ref_frames(ref, pair)
    img_gen = ref.dr.estimate_depth_kf_native(src2, T(pair.kld_gt), T(pair.pose_gt))
    # constant depth for all segments -> exact half pixel shift
    kldc = np.full(N, np.log(2.0), np.float32)
    pose_h = np.eye(4, dtype=np.float32)
    pose_h[0, 3] = 0.5 * 2.0 / pair.K[0, 0]
    pose_h[1, 3] = 0.5 * 2.0 / pair.K[1, 1]
    img_half = ref.dr.estimate_depth_kf_native(src, T(kldc), T(pose_h))
    save = pair_inputs(pair)
    # mean=True (core/ops.py:84-92): scatter_reduce 'mean' incl. the initial zero; a pose that piles several points onto a pixel
    pose_far = np.eye(4, dtype=np.float32)
    pose_far[2, 3] = 2.5
    img_mean = ref.dr.estimate_depth_kf_native(src2, T(pair.kld_gt), T(pose_far), mean=True)
    img_mean_gen = ref.dr.estimate_depth_kf_native(src2, T(pair.kld_gt), T(pair.pose_gt), mean=True)
    save.update(in_L_const=L, in_kld_levels=kld, out_identity=img_id.numpy(), in_kld_gt=pair.kld_gt,
                in_pose_gt=pair.pose_gt, out_general=img_gen.numpy(), in_kld_const=kldc, in_pose_half=pose_h,
                out_half=img_half.numpy(), in_pose_far=pose_far, out_mean_far=img_mean.numpy(), out_mean_general=img_mean_gen.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_segment_stats(ref, name):
    """G7: segment_based_depth_reinit (mean, median; one invisible segment; even counts) and
    unproject_kf_to_depths + the per-pixel average used by depth completion."""
    pair = synth.make_pair(50, 70, 7, seed=21, shape="blobs")
    src, _ = ref_frames(ref, pair)
    rng = np.random.default_rng(9)
    est = pair.depth * np.exp(0.05 * rng.standard_normal(pair.depth.shape)).astype(np.float32)
    sparse = np.where(rng.uniform(size=est.shape) < 0.08, est, 0.0).astype(np.float32)
    sparse[pair.keypoint_regions[2]] = 0.0          # segment 2 sees no measurement
    save = pair_inputs(pair)
    save["in_sparse_depth"] = sparse
    for mode in ("mean", "median"):
        kld, vis = ref.di.segment_based_depth_reinit(T(sparse).clone(), src, mode=mode, return_info=True)
        torch.set_grad_enabled(True)
        save[f"{mode}_kld"] = kld.numpy()
        save[f"{mode}_visible"] = vis.numpy()
    kld, vis = T(save["median_kld"]), torch.from_numpy(save["median_visible"])
    depths = ref.do.unproject_kf_to_depths(src, kld)
    save["depths_dense"] = depths.numpy().astype(np.float16).astype(np.float32)  # size: coarse copy, exact one below
    save["depths_dense_sum"] = depths.double().sum().numpy()
    d = depths.clone()
    d[src.keypoint_regions == 0] = -1
    d = d[vis]
    # render_depth_avg (depth_completion/segment_based_completion.py:21-27), restated inline because that
    # module imports the SAM frontend and cannot be imported here
    invalid = d.max(dim=0)[0] < 1e-6
    d[d < 1e-6] = 0.0
    avg = d.sum(dim=0) / ((d > 1e-6).sum(dim=0) + 1e-6)
    save["avg_depth"] = avg.numpy()
    save["avg_invalid"] = invalid.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_lie(ref, name):
    """G8: renormalise_se3, invertSE3, torch_pose_to_tq, SE3_logmap, quaternion_to_matrix."""
    rng = np.random.default_rng(2)
    Ts = np.stack([synth.se3_exp_np(rng.standard_normal(6) * s) for s in (0.01, 0.3, 1.0, 2.5, 3.0)]).astype(np.float32)
    noisy = Ts.copy()
    noisy[:, :3, :3] += (1e-3 * rng.standard_normal((5, 3, 3))).astype(np.float32)
    save = dict(in_T=Ts, in_noisy=noisy)
    save["renorm"] = ref.la.renormalise_se3(T(noisy).clone()).numpy()
    save["inverse"] = ref.la.invertSE3(T(Ts)).numpy()
    save["tq"] = ref.la.torch_pose_to_tq(T(Ts)).numpy()
    save["tq_single"] = ref.la.torch_pose_to_tq(T(Ts[1])).numpy()
    save["logmap"] = np.stack([ref.la.SE3_logmap(T(Ts[i:i + 1])).numpy()[0] for i in range(4)])  # only B=1 broadcasts correctly upstream
    q = rng.standard_normal((6, 4)).astype(np.float32)
    save["in_quat"] = q
    save["quat_R"] = ref.la.quaternion_to_matrix(T(q)).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


# ----------------------------------------------------------------------------
# G9: K-step Adam trajectories of the three caller loop shapes.  The loops are
# restated here (the reference drivers import SAM/cupy and hard-code cuda:0) around the REAL reference cost
# functions; the pose parameterisation uses the build's SE(3) exp since lietorch is absent (parity unpinned
# at that boundary).
# ----------------------------------------------------------------------------
def golden_traj_sfm(ref, name, steps=40):
    pair = synth.make_pair(48, 64, 6, seed=31, init_sigma=0.01)
    src, trg = ref_frames(ref, pair)
    sp = ref.kf.keyframe_pyramid(src, 0, 2)
    tp = ref.kf.keyframe_pyramid(trg, 0, 2)
    kld = torch.nn.Parameter(T(pair.kld_init))
    a = torch.nn.Parameter(torch.zeros(1, 6))
    T0 = T(pair.pose_init)
    opt = torch.optim.Adam([{"params": kld, "lr": 1e-3}, {"params": [a], "lr": 1e-2}], lr=1e-3)
    losses, count = [], 0
    for s, t in zip(sp, tp):
        for _ in range(steps):
            pose = orc.se3_exp(a)[0] @ T0
            out = ref.do.photomeric_cost(s, t, kld, pose, {"mode": "colour", "collect_stats": 0})
            loss = torch.sum(torch.stack([torch.mean(torch.abs(out["residual"]))]))
            losses.append(float(loss))
            if count > 0:
                loss.backward()
                opt.step()
                opt.zero_grad()
            count += 1
    save = pair_inputs(pair)
    save.update(in_pose_init=pair.pose_init, in_kld=pair.kld_init, steps=np.array(steps), losses=np.array(losses),
                final_kld=kld.detach().numpy(), final_a=a.detach().numpy(),
                final_pose=(orc.se3_exp(a.detach())[0] @ T0).numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_converged(ref, name, steps=350):
    """The minimiser of the REAL reference cost, for the Gauss-Newton/LM solver (which the reference does not have) to be
    pinned against: the two-frame SfM loop of golden_traj_sfm run to convergence over 3 levels, then polished at the
    finest level with the learning rates divided by 10 and by 100 so that Adam's fixed-step jitter is below 1e-5."""
    pair = synth.make_pair(60, 80, 8, seed=41, init_sigma=0.01)
    src, trg = ref_frames(ref, pair)
    sp = ref.kf.keyframe_pyramid(src, 0, 3)
    tp = ref.kf.keyframe_pyramid(trg, 0, 3)
    kld = torch.nn.Parameter(T(pair.kld_init))
    a = torch.nn.Parameter(torch.zeros(1, 6))
    T0 = T(pair.pose_init)
    cfg = {"mode": "colour", "collect_stats": 0}
    losses = []

    def run(s, t, n, scale):
        opt = torch.optim.Adam([{"params": kld, "lr": 1e-3 * scale}, {"params": [a], "lr": 1e-2 * scale}], lr=1e-3)
        for _ in range(n):
            pose = orc.se3_exp(a)[0] @ T0
            loss = torch.mean(torch.abs(ref.do.photomeric_cost(s, t, kld, pose, cfg)["residual"]))
            losses.append(float(loss))
            loss.backward()
            opt.step()
            opt.zero_grad()

    for s, t in zip(sp, tp):
        run(s, t, steps, 1.0)
    for scale in (0.1, 0.01):
        run(sp[-1], tp[-1], steps, scale)
    with torch.no_grad():
        pose = orc.se3_exp(a)[0] @ T0
        final = float(torch.mean(torch.abs(ref.do.photomeric_cost(sp[-1], tp[-1], kld, pose, cfg)["residual"])))
    save = pair_inputs(pair)
    save.update(in_pose_init=pair.pose_init, in_kld=pair.kld_init, losses=np.array(losses), final_loss=np.array(final),
                final_kld=kld.detach().numpy(), final_pose=pose.numpy(), pose_gt=pair.pose_gt, kld_gt=pair.kld_gt)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_traj_track(ref, name, steps=40):
    pair = synth.make_pair(48, 64, 6, seed=32, init_sigma=0.01)
    src, trg = ref_frames(ref, pair)
    with torch.no_grad():
        pre = ref.do.unproject_kf(src, T(pair.kld_gt))
    delta = torch.nn.Parameter(torch.zeros(1, 6))
    aff = torch.nn.Parameter(torch.zeros(2))
    prev_aff = torch.zeros(2)
    opt = torch.optim.Adam([{"params": [delta], "lr": 5e-3}, {"params": [aff], "lr": 5e-3}], lr=5e-3)
    prev_pose = torch.eye(4)
    supp_T = ref.la.invertSE3(T(pair.pose_init))          # pose = inv(supp_T) @ prev_pose = pose_init
    losses = []
    for _ in range(steps):
        pose = orc.se3_exp(delta)[0] @ ref.la.invertSE3(supp_T) @ prev_pose
        out = ref.do.photomeric_cost_precomputed(pre, trg, pose, {"mode": "colour", "collect_stats": 0},
                                                 affine_comp=(prev_aff, aff))
        loss = torch.mean(out["residual"])
        losses.append(float(loss))
        loss.backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            supp_T = supp_T @ ref.la.invertSE3(orc.se3_exp(delta.detach())[0])
            delta.data = torch.zeros_like(delta.data)
    supp_T = ref.la.renormalise_se3(supp_T)
    save = pair_inputs(pair)
    save.update(in_pose_init=pair.pose_init, in_kld=pair.kld_gt, steps=np.array(steps), losses=np.array(losses),
                final_supp_T=supp_T.numpy(), final_aff=aff.detach().numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_traj_map(ref, name, steps=30):
    """One source KF against B=2 targets; kld + per-target delta poses + affines; fold-in + renormalise."""
    pair = synth.make_pair(48, 64, 6, seed=33, init_sigma=0.01)
    other = synth.make_pair(48, 64, 6, seed=33, init_sigma=0.01, motion_scale=1.8)
    src, _ = ref_frames(ref, pair)
    imgs = T(np.stack([pair.trg_image, other.trg_image]))
    Ks = T(np.stack([pair.K, pair.K]))
    poses = [T(pair.pose_init), T(other.pose_init)]        # T_trg<-src estimates, refined in place
    kld = torch.nn.Parameter(T(pair.kld_init))
    deltas = [torch.nn.Parameter(torch.zeros(1, 6)) for _ in range(2)]
    affs = [torch.nn.Parameter(torch.zeros(2)) for _ in range(2)]
    aff_src = torch.zeros(2)
    opt = torch.optim.Adam([{"params": kld, "lr": 1e-2}, {"params": deltas, "lr": 1e-2}, {"params": affs, "lr": 1e-5}],
                           lr=1e-3)
    losses = []
    for _ in range(steps):
        P = torch.stack([orc.se3_exp(d)[0] @ p for d, p in zip(deltas, poses)])
        out = ref.dob.photomeric_cost_batch(src, imgs, Ks, kld, P, {"mode": "colour", "collect_stats": 0},
                                            affine_comp=(aff_src, torch.stack(affs)))
        loss = torch.mean(out["residual"])
        losses.append(float(loss))
        loss.backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            for i in range(2):
                poses[i] = ref.la.renormalise_se3(orc.se3_exp(deltas[i].detach())[0] @ poses[i])
                deltas[i].data = torch.zeros_like(deltas[i].data)
    save = pair_inputs(pair)
    save.update(in_trg_images=imgs.numpy(), in_poses_init=np.stack([pair.pose_init, other.pose_init]),
                in_kld=pair.kld_init, steps=np.array(steps), losses=np.array(losses), final_kld=kld.detach().numpy(),
                final_poses=torch.stack(poses).numpy(), final_affs=torch.stack([a.detach() for a in affs]).numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def reference_mapping_loop(ref, kfs, kf_poses, kf_klds, kf_affs, supp, steps, lr_pose, window_size, affine, initialised, lr_kld=1e-2, lr_aff=1e-5, mode="map"):
    """The windowed mapping loop of odometery/odometery.py:576-648 (parameter groups), :451-479 (neighbour connectivity),
    :756-915 (iteration) restated around the REAL ``photomeric_cost_batch`` / ``renormalise_se3`` with ``opt_supporting``
    on; lietorch's Exp is replaced by the oracle's (parity unpinned at that boundary).
    kfs: reference KeyFrames; kf_poses: camera-to-world (4,4); supp[k] = list of (KeyFrame, pose, affine).
    mode 'supp' (the supplementary mapping after every tracked frame, :1038-1042): only the latest keyframe is a source (:467-469), the
    optimiser holds only ITS log-depths (:616-619; no pose group :586-588, no affine groups :628-629,634-635, supporting deltas are
    plain identities :544-560)."""
    K = len(kfs)
    eye6 = lambda: torch.zeros(1, 6)
    kf_poses = [p.clone() for p in kf_poses]
    supp_mode = mode == "supp"
    d_kf = [None] + [None if supp_mode else torch.nn.Parameter(eye6()) for _ in range(K - 1)]       # first keyframe fixed (:589-592)
    if supp_mode:
        klds = [k.clone() for k in kf_klds[:-1]] + [torch.nn.Parameter(kf_klds[-1].clone())]
    elif K == window_size:                                                                    # oldest depths frozen (:594-603)
        klds = [kf_klds[0].clone()] + [torch.nn.Parameter(k.clone()) for k in kf_klds[1:]]
    else:
        klds = [torch.nn.Parameter(k.clone()) for k in kf_klds]
    par = (lambda x: x.clone()) if supp_mode else (lambda x: torch.nn.Parameter(x.clone()))
    affs = [kf_affs[0].clone()] + [par(a) for a in kf_affs[1:]] if affine else [None] * K
    s_pose = [[p.clone() for _, p, _ in supp[k]] for k in range(K)]
    s_delta = [[None if supp_mode else torch.nn.Parameter(eye6()) for _ in supp[k]] for k in range(K)]
    s_aff = [[par(a) for _, _, a in supp[k]] for k in range(K)] if affine else None
    groups = [{"params": [k for k in klds if isinstance(k, torch.nn.Parameter)], "lr": lr_kld},
              {"params": [d for d in d_kf if d is not None], "lr": lr_pose}]
    if not supp_mode:
        if affine:
            groups.append({"params": affs[1:], "lr": lr_aff})
        groups.append({"params": [d for row in s_delta for d in row], "lr": lr_pose})
        if affine:
            groups.append({"params": [a for row in s_aff for a in row], "lr": lr_aff})
    opt = torch.optim.Adam(groups, lr=1e-3)
    mat = lambda d: torch.eye(4) if d is None else orc.se3_exp(d)[0]
    conn = {s: [t for t in (s - 1, s + 1) if 0 <= t < K] for s in range(K) if not (supp_mode and s != K - 1)}
    cfg = {"mode": "colour", "collect_stats": 0}
    losses, prev, stopped = [], float("inf"), -1
    for it in range(steps):
        res = []
        for s, trgs in conn.items():
            src_delta = mat(d_kf[s])
            imgs, Ks, Ps, As = [], [], [], []
            for t in trgs:
                imgs.append(kfs[t].image); Ks.append(kfs[t].K)
                Ps.append(mat(d_kf[t]) @ torch.linalg.inv(kf_poses[t]) @ kf_poses[s] @ torch.linalg.inv(src_delta))
                As.append(affs[t])
            for ss in ([s] + ([s - 1] if s > 0 else [])):
                for j, (f, _, _) in enumerate(supp[ss]):
                    imgs.append(f.image); Ks.append(f.K)
                    Ps.append(mat(s_delta[ss][j]) @ torch.linalg.inv(s_pose[ss][j]) @ kf_poses[s] @ torch.linalg.inv(src_delta))
                    As.append(s_aff[ss][j] if affine else None)
            out = ref.dob.photomeric_cost_batch(kfs[s], torch.stack(imgs), torch.stack(Ks), klds[s], poses=torch.stack(Ps),
                                                affine_comp=(affs[s], torch.stack(As)) if affine else None, cost_config=cfg)
            res.append(out["residual"].mean())
        loss = torch.sum(torch.stack(res))
        losses.append(float(loss.detach()))
        loss.backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            for i in range(K):
                kf_poses[i] = ref.la.renormalise_se3(kf_poses[i] @ torch.linalg.inv(mat(d_kf[i])))
                if d_kf[i] is not None:
                    d_kf[i].data.zero_()
            for k in range(K):
                for j in range(len(s_pose[k])):
                    s_pose[k][j] = ref.la.renormalise_se3(s_pose[k][j] @ torch.linalg.inv(mat(s_delta[k][j])))
                    if s_delta[k][j] is not None:
                        s_delta[k][j].data.zero_()
        if initialised:
            if abs(losses[-1] - prev) / prev < 1e-8:
                stopped = it
                break
            prev = losses[-1]
    det = lambda x: x.detach().clone()
    if len({int(k.shape[0]) for k in klds}) > 1:            # (keyframes of a sequence differ in their segment count)
        klds_out = np.array([det(k).numpy() for k in klds], dtype=object)
    else:
        klds_out = torch.stack([det(k) for k in klds]).numpy()
    return dict(losses=np.array(losses), stopped=np.array(stopped), kf_poses=torch.stack(kf_poses).numpy(),
                klds=klds_out,
                affs=torch.stack([det(a) for a in affs]).numpy() if affine else np.zeros((K, 2), np.float32),
                supp_poses=torch.stack([p for row in s_pose for p in row]).numpy() if any(s_pose) else np.zeros((0, 4, 4), np.float32),
                supp_affs=torch.stack([det(a) for row in s_aff for a in row]).numpy() if affine and any(s_aff) else np.zeros((0, 2), np.float32))


def golden_traj_window(ref, name, steps=30):
    """G9-d: multi-source windowed mapping.  Case 'full': 3 keyframes = window size (oldest depths frozen, first pose fixed),
    map-mode learning rates, early stop armed.  Case 'init': 2 keyframes, window not full, mono-init pose rate 1e-2."""
    save = {}
    for tag, (n_kf, lr_pose, window, initialised, seed) in {"full": (3, 1e-4, 3, True, 35), "init": (2, 1e-2, 5, False, 36)}.items():
        frames, est, klds, affs = synth.window_inputs(seed, n_kf)
        kfs = [ref.kf.KeyFrame(T(f.image), T(f.K), T(f.logdepth_perseg), T(f.keypoints), torch.from_numpy(f.keypoint_regions.copy()))
               for f in frames[0::2]]
        supp = [[(ref.kf.KeyFrame(T(frames[2 * k + 1].image), T(frames[2 * k + 1].K)), T(est[2 * k + 1]), T(affs[2 * k + 1]))]
                for k in range(n_kf)]
        out = reference_mapping_loop(ref, kfs, [T(est[2 * k]) for k in range(n_kf)], [T(k) for k in klds],
                                     [T(affs[2 * k]) for k in range(n_kf)], supp, steps, lr_pose, window, True, initialised)
        save.update({f"{tag}_{k}": v for k, v in out.items()})
        save.update({f"{tag}_cfg": np.array([n_kf, window, int(initialised), seed, steps]), f"{tag}_lr_pose": np.array(lr_pose),
                     f"{tag}_in_poses": np.stack(est), f"{tag}_in_klds": np.stack(klds), f"{tag}_in_affs": affs})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_traj_supp(ref, name, steps=12):
    """G9-e: the SUPPLEMENTARY mapping (odometery.py:1038-1042, mode 'supp'): 3 keyframes with one supporting frame each + two running
    ones for the latest; only the latest keyframe is a source, only its log-depths are optimised, every pose and affine pair stays."""
    frames, kfi, si, est, klds, affs = synth.reference_window_inputs(37, 3, 1, 2, H=48, W=64, N=6)
    kfs = [ref.kf.KeyFrame(T(frames[i].image), T(frames[i].K), T(frames[i].logdepth_perseg), T(frames[i].keypoints),
                           torch.from_numpy(frames[i].keypoint_regions.copy())) for i in kfi]
    supp = [[(ref.kf.KeyFrame(T(frames[j].image), T(frames[j].K)), T(est[j]), T(affs[j])) for j in row] for row in si]
    out = reference_mapping_loop(ref, kfs, [T(est[i]) for i in kfi], [T(k) for k in klds], [T(affs[i]) for i in kfi], supp, steps, 1e-4, 5,
                                 True, True, mode="supp")
    save = {f"supp_{k}": v for k, v in out.items()}
    save.update(cfg=np.array([37, 3, 1, 2, 48, 64, 6, steps]), in_poses=np.stack(est), in_klds=np.stack(klds), in_affs=affs)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_post_process(ref, name):
    """N2: depth_discontinuity, mask_by_depth_discontinuity, connected_components_batch and
    kf_fix_disconnected_regions (torch seed 123 for the re-seeded keypoints) on keyframes with depth steps."""
    save = {}
    for tag, (H, W, N, shape, seed) in {"grid": (60, 80, 6, "grid", 5), "blobs": (72, 96, 9, "blobs", 6)}.items():
        pair = synth.make_pair(H, W, N, seed=seed, shape=shape)
        L = synth.stepped_logdepth(pair, seed=seed)
        kf = ref.kf.KeyFrame(T(pair.src_image), T(pair.K), T(L), T(pair.keypoints), torch.from_numpy(pair.keypoint_regions.copy()))
        disc = ref.pp.depth_discontinuity(T(L), torch.from_numpy(pair.keypoint_regions.copy()))
        split = ref.pp.mask_by_depth_discontinuity(T(L), torch.from_numpy(pair.keypoint_regions.copy()))
        torch.set_grad_enabled(True)
        lab, n_lab = ref.pp.connected_components_batch(split.numpy())
        torch.manual_seed(123)
        new = ref.pp.kf_fix_disconnected_regions(kf)
        torch.set_grad_enabled(True)
        save.update({f"{tag}_HWN": np.array([H, W, N]), f"{tag}_L": L, f"{tag}_keypoints": pair.keypoints,
                     f"{tag}_masks": np.packbits(pair.keypoint_regions, axis=-1), f"{tag}_disc": np.packbits(disc.numpy(), axis=-1),
                     f"{tag}_split": np.packbits(split.numpy(), axis=-1), f"{tag}_labels": lab.astype(np.int32),
                     f"{tag}_n_labels": np.array(n_lab), f"{tag}_new_masks": np.packbits(new.keypoint_regions.numpy(), axis=-1),
                     f"{tag}_new_K": np.array(new.keypoint_regions.shape[0]), f"{tag}_new_logdepth_sum": new.logdepth_perseg.double().sum((1, 2)).numpy(),
                     f"{tag}_new_keypoints": new.keypoints.numpy()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_kf_criteria(ref, name):
    """N3: translation_difference / rotation_difference on rendered-depth-like images: holes (zeros), even and odd valid
    counts, a single valid pixel, large and tiny rotations."""
    rng = np.random.default_rng(2024)
    save = {}
    cases = {"odd": (37, 53, 0.3), "even": (48, 64, 0.25), "dense": (60, 80, 0.0), "one": (8, 9, None)}
    for tag, (H, W, holes) in cases.items():
        depth = rng.uniform(0.4, 6.0, (H, W)).astype(np.float32)
        if holes is None:
            depth[:] = 0.0
            depth[3, 4] = 2.5
        else:
            depth[rng.uniform(size=(H, W)) < holes] = 0.0
            if tag == "even" and (depth > 1e-6).sum() % 2:
                depth[np.argwhere(depth > 1e-6)[0][0], np.argwhere(depth > 1e-6)[0][1]] = 0.0
            if tag == "odd" and (depth > 1e-6).sum() % 2 == 0:
                depth[np.argwhere(depth > 1e-6)[0][0], np.argwhere(depth > 1e-6)[0][1]] = 0.0
        scale_rot = {"odd": 0.05, "even": 1.2, "dense": 1e-4, "one": 2.9}[tag]
        a = synth.se3_exp_np(np.concatenate([rng.standard_normal(3), 0.3 * rng.standard_normal(3)])).astype(np.float32)
        b = (synth.se3_exp_np(np.concatenate([0.4 * rng.standard_normal(3), scale_rot * np.array([0.6, -0.48, 0.64])]))
             @ a.astype(np.float64)).astype(np.float32)
        diff, scale = ref.kc.translation_difference(T(a), T(b), T(depth))
        ang = ref.kc.rotation_difference(T(a), T(b))
        save.update({f"{tag}_depth": depth, f"{tag}_pose_src": a, f"{tag}_pose_trg": b, f"{tag}_diff": np.float32(diff.item()),
                     f"{tag}_scale": np.float32(scale.item()), f"{tag}_angle_deg": np.float64(ang),
                     f"{tag}_n_valid": np.int64((depth > 1e-6).sum())})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def golden_helpers(ref, name):
    """Thin helpers whose mirrors are plain tensor expressions: depth pyramid steps in every mode, the depth / intrinsics
    pyramid modules' level selection, normal-channel rotation, host-conversion helpers."""
    rng = np.random.default_rng(99)
    depth = rng.uniform(0.5, 4.0, (2, 1, 9, 12)).astype(np.float32)
    holes = depth.copy()
    holes[rng.uniform(size=holes.shape) < 0.4] = np.nan
    holes[0, 0, :2, :2] = np.nan
    save = {"depth": depth, "holes": holes}
    for mode in ("bilinear", "nearest_neighbor", "max", "min"):
        save[f"pyr_{mode}"] = ref.gp.pyr_depth(T(depth), mode, 2).numpy()
    save["pyr_masked_bilinear"] = ref.gp.pyr_depth(T(holes), "masked_bilinear", 2).numpy()
    for (s0, e0) in ((0, 3), (1, 4), (2, 3), (0, 1)):
        lv = ref.gp.DepthPyramidModule(s0, e0, "nearest_neighbor", "cpu")(T(np.tile(depth, (1, 1, 4, 4))))
        save[f"dpyr_{s0}_{e0}_n"] = np.array(len(lv))
        for i, l in enumerate(lv):
            save[f"dpyr_{s0}_{e0}_{i}"] = l.numpy()
        Ks = ref.gp.IntrinsicsPyramidModule(s0, e0, "cpu")(T(np.array([[500., 0, 320], [0, 510., 240], [0, 0, 1]], np.float32)), [1.0, 0.5])
        save[f"kpyr_{s0}_{e0}"] = torch.stack(Ks).numpy()
    px = rng.standard_normal((1, 7, 11)).astype(np.float32)
    poses = np.stack([synth.se3_exp_np(0.3 * rng.standard_normal(6)) for _ in range(3)]).astype(np.float32)
    save.update(px=px, poses=poses)
    for mode, C in (("colour", 3), ("colour_norm", 6), ("colour_norm_kappa", 7)):
        save[f"nrm_{mode}"] = ref.nc.transform_normals_batch(T(px[:, :C]), T(poses), mode).numpy()
        save[f"nrm1_{mode}"] = ref.nc.transform_normals(T(px[:, :C]), T(poses[1]), mode).numpy()
    img = rng.uniform(0, 1, (3, 5, 7)).astype(np.float32)
    u8 = (rng.uniform(0, 255, (5, 7, 3))).astype(np.uint8)
    save.update(img=img, u8=u8, to_img=ref.etc.to_img(T(img)), to_img_np=ref.etc.to_img_np(T(img)),
                image_tt=ref.etc.image_tt(u8, device="cpu").numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    if len(sys.argv) > 1:                      # regenerate selected fixtures only: python oracle/gen_goldens.py g9d_traj_window
        for name in sys.argv[1:]:
            {"g9d_traj_window": golden_traj_window, "g9e_traj_supp": golden_traj_supp}[name](ref, name)
        return

    p = synth.make_pair(48, 64, 6, seed=1)
    golden_cost(ref, "g1_grid_48x64", p, p.pose_init)

    p = synth.make_pair(60, 80, 8, seed=2, shape="blobs")
    rng = np.random.default_rng(77)
    pose = (synth.se3_exp_np(0.05 * rng.standard_normal(6)) @ p.pose_gt.astype(np.float64)).astype(np.float32)
    golden_cost(ref, "g1_blobs_affine_60x80", p, pose,
                affine=(np.array([0.02, -0.015], np.float32), np.array([-0.04, 0.03], np.float32)))

    p = synth.make_pair(72, 96, 12, seed=3, overlap=3)
    golden_cost(ref, "g1_pyramid_72x96", p, p.pose_init, levels=(0, 3))

    p = synth.make_pair(48, 64, 6, seed=4)
    golden_cost(ref, "g1_behind_camera_48x64", p, big_rotation_pose(p, 1.4))

    p = synth.make_pair(45, 67, 5, seed=5, drop_border=2)
    golden_cost(ref, "g1_odd_45x67", p, p.pose_init, levels=(1, 3),
                affine=(np.array([0.0, 0.0], np.float32), np.array([0.1, -0.05], np.float32)))

    p = synth.make_pair(60, 80, 8, seed=6, shape="blobs")
    golden_precomputed(ref, "g3_precomputed_60x80", p, p.pose_init,
                       (np.array([0.01, 0.02], np.float32), np.array([-0.03, 0.01], np.float32)))

    p = synth.make_pair(48, 64, 6, seed=7, overlap=2)
    golden_batch(ref, "g4_batch3_48x64", p)

    golden_pyramid(ref, "g5_pyramid")
    golden_depth_render(ref, "g6_depth_render")
    golden_segment_stats(ref, "g7_segment_stats")
    golden_lie(ref, "g8_lie")
    golden_traj_sfm(ref, "g9a_traj_sfm")
    golden_traj_track(ref, "g9b_traj_track")
    golden_traj_map(ref, "g9c_traj_map")
    golden_post_process(ref, "g10_post_process")
    golden_kf_criteria(ref, "g11_kf_criteria")
    golden_converged(ref, "g12_converged_sfm")
    golden_helpers(ref, "g13_helpers")
    golden_traj_window(ref, "g9d_traj_window")
    golden_traj_supp(ref, "g9e_traj_supp")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
