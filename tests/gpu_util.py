"""Helpers for the -m gpu tests: build cuda keyframes from golden fixtures / synthetic pairs."""
import numpy as np
import torch

from conftest import unpack_masks


def dev():
    return torch.device("cuda:0")


def T(a, requires_grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t.requires_grad_(True) if requires_grad else t


def frames_from_golden(g, src_img=None, trg_img=None, K_img=None):
    from super_primitive_amd.image.keyframe import KeyFrame
    masks = unpack_masks(g)
    src = KeyFrame(T(g["in_src_image"] if src_img is None else src_img), T(g["in_K"]), T(g["in_logdepth"]),
                   T(g["in_keypoints"]), T(masks), K_img=None if K_img is None else T(K_img))
    trg = KeyFrame(T(g["in_trg_image"] if trg_img is None else trg_img), T(g["in_K"]),
                   K_img=None if K_img is None else T(K_img))
    return src, trg


def frames_from_synth(pair):
    from super_primitive_amd.image.keyframe import KeyFrame
    src = KeyFrame(T(pair.src_image), T(pair.K), T(pair.logdepth_perseg), T(pair.keypoints), T(pair.keypoint_regions))
    trg = KeyFrame(T(pair.trg_image), T(pair.K))
    return src, trg


def npy(t):
    return t.detach().cpu().numpy()


def assert_masks_close(got, want, max_flips=2, what="mask"):
    got, want = np.asarray(got).astype(bool), np.asarray(want).astype(bool)
    assert got.shape == want.shape, what
    flips = int((got != want).sum())
    assert flips <= max_flips, f"{what}: {flips} entries differ"
    return got == want
