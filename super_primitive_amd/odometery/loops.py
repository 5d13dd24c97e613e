"""The two inner-loop shapes of the reference's MonoVO driver (``odometery/odometery.py``) on the HIP cost:
frame-to-keyframe tracking (``:300-312,323-407``) and windowed mapping of one source keyframe against a batch of
targets (``:576-648,687-915``).  Keyframe management, supporting-frame bookkeeping, GUI queues and checkpoints
stay a driver concern and are out of scope; these functions are the part that sits on the hot path
and are what BASELINE config 3 exercises."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..core import dense_optim, dense_optim_batch
from ..lie.lie_algebra import invertSE3, renormalise_se3
from ..lie.lietorch_utils import zero_out_lietorch_tensor
from ..lie.se3 import SE3, LieGroupParameter, se3_exp_matrix

CFG = {'mode': 'colour', 'collect_stats': 0}


def track_frame(kf_precomputed_levels, supp_levels, supp_T, prev_pose, steps, lr=5e-3, prev_aff=None, curr_aff=None):
    """Adam on a zero-reset pose tangent (+ affine) against precomputed source points
    (odometery.py:300-312 setup_tracking_opt, :375-407 the loop).

    kf_precomputed_levels / supp_levels: per pyramid level (coarse -> fine) the ``unproject_kf`` dict and the
    supporting KeyFrame; steps: iterations per level (config ``track.steps``).  Returns (supp_T, curr_aff, losses)."""
    dev = supp_T.device
    delta = LieGroupParameter(SE3.Identity(1, device=dev))
    params = [{'params': [delta], 'lr': lr}]
    affine = prev_aff is not None
    if affine:
        curr_aff = nn.Parameter(curr_aff.detach().clone())
        params.append({'params': [curr_aff], 'lr': 5e-3})
    optim = torch.optim.Adam(params, lr=5e-3)
    supp_T = supp_T.detach().clone()
    losses = []
    for level, n_steps in enumerate(steps):
        for _ in range(n_steps):
            pose = delta.retr().matrix()[0] @ invertSE3(supp_T) @ prev_pose
            aff = (prev_aff, curr_aff) if affine else None
            out = dense_optim.photomeric_cost_precomputed(kf_precomputed_levels[level], supp_levels[level], pose, CFG,
                                                          affine_comp=aff)
            loss = torch.mean(out['residual'])
            losses.append(loss.detach())
            loss.backward()
            optim.step()
            optim.zero_grad()
            with torch.no_grad():
                supp_T = supp_T @ invertSE3(se3_exp_matrix(delta.detach().as_subclass(torch.Tensor))[0])
                zero_out_lietorch_tensor(delta)          # Adam moments persist (odometery.py:400-403)
    supp_T = renormalise_se3(supp_T.contiguous())
    return supp_T, (curr_aff.detach() if affine else None), losses


def map_source_against_targets(src_kf, trg_images, trg_Ks, kld, poses, steps, lr_kld=1e-2, lr_pose=1e-2, lr_aff=1e-5,
                               aff_src=None, affs=None, rel_tol=1e-8):
    """One source keyframe, B targets: Adam on kld + per-target delta poses + affines with fold-in,
    renormalisation and tangent reset every iteration, relative-loss early stop (odometery.py:756-915 restricted
    to a single source keyframe).  Returns (kld, poses (B,4,4), affs, losses)."""
    dev = kld.device
    B = poses.shape[0]
    kld = nn.Parameter(kld.detach().clone())
    deltas = [LieGroupParameter(SE3.Identity(1, device=dev)) for _ in range(B)]
    poses = [p.detach().clone() for p in poses]
    affine = affs is not None
    groups = [{'params': [kld], 'lr': lr_kld}, {'params': deltas, 'lr': lr_pose}]
    if affine:
        affs = [nn.Parameter(a.detach().clone()) for a in affs]
        groups.append({'params': affs, 'lr': lr_aff})
    optim = torch.optim.Adam(groups, lr=1e-3)
    losses = []
    prev = None
    for _ in range(steps):
        P = torch.stack([d.retr().matrix()[0] @ p for d, p in zip(deltas, poses)])
        aff = (aff_src, torch.stack(list(affs))) if affine else None
        out = dense_optim_batch.photomeric_cost_batch(src_kf, trg_images, trg_Ks, kld, P, CFG, affine_comp=aff)
        loss = torch.mean(out['residual'])
        losses.append(loss.detach())
        loss.backward()
        optim.step()
        optim.zero_grad()
        with torch.no_grad():
            for i in range(B):
                step = se3_exp_matrix(deltas[i].detach().as_subclass(torch.Tensor))[0]
                poses[i] = renormalise_se3((step @ poses[i]).contiguous())
                zero_out_lietorch_tensor(deltas[i])
        cur = float(losses[-1])
        if prev is not None and abs(prev - cur) / max(abs(prev), 1e-30) < rel_tol:
            break
        prev = cur
    return kld.detach(), torch.stack(poses), ([a.detach() for a in affs] if affine else None), losses
