"""-m gpu: pyramid, depth render, segment statistics, dense depth expansion and SE(3) helpers on the HIP path
against the reference's golden vectors (and the oracle where the fixture carries no output)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, unpack_masks
from gpu_util import T, frames_from_golden, npy

pytestmark = pytest.mark.gpu


def test_keyframe_pyramid_matches_reference():
    from super_primitive_amd.image.keyframe import KeyFrame, keyframe_pyramid
    g = load_golden("g5_pyramid")
    for tag in ("even", "odd"):
        kf = KeyFrame(T(g[f"{tag}_image"]), T(g[f"{tag}_Kin"]))
        for (s, e) in ((0, 3), (1, 4), (0, 1)):
            pyr = keyframe_pyramid(kf, s, e)
            assert len(pyr) == int(g[f"{tag}_{s}_{e}_n"])
            for i, k in enumerate(pyr):
                want = g[f"{tag}_{s}_{e}_img{i}"]
                assert tuple(k.image.shape) == want.shape
                np.testing.assert_allclose(npy(k.image), want, rtol=0, atol=3e-7)
                np.testing.assert_allclose(npy(k.K_img), g[f"{tag}_{s}_{e}_Kimg{i}"], rtol=1e-6)
                np.testing.assert_allclose(npy(k.K), g[f"{tag}_{s}_{e}_K{i}"], rtol=0)
                assert k.is_supporting()


def test_depth_render_matches_reference():
    from super_primitive_amd.core.depth_render import estimate_depth_kf_native
    from super_primitive_amd.image.keyframe import KeyFrame
    g = load_golden("g6_depth_render")
    masks = unpack_masks(g)
    kf = KeyFrame(T(g["in_src_image"]), T(g["in_K"]), T(g["in_L_const"]), T(g["in_keypoints"]), T(masks))
    half = estimate_depth_kf_native(kf, T(g["in_kld_const"]), T(g["in_pose_half"]))
    np.testing.assert_allclose(npy(half), g["out_half"], rtol=1e-6)          # collision-free, unambiguous truncation
    ident = estimate_depth_kf_native(kf, T(g["in_kld_levels"]))
    # identity pose: every point lands on u = c +- 1e-5, so the truncation (v,u).long() of the reference is decided
    # by the last bits of exp() / division and differs between ATen-CPU and the GPU for a few percent of pixels
    assert np.isclose(npy(ident), g["out_identity"], rtol=1e-6).mean() > 0.95
    kf2 = KeyFrame(T(g["in_src_image"]), T(g["in_K"]), T(g["in_logdepth"]), T(g["in_keypoints"]), T(masks))
    gen = npy(estimate_depth_kf_native(kf2, T(g["in_kld_gt"]), T(g["in_pose_gt"])))
    assert np.isclose(gen, g["out_general"], rtol=1e-5).mean() > 0.98
    assert np.array_equal(gen > 0, g["out_general"] > 0) or ((gen > 0) != (g["out_general"] > 0)).mean() < 0.01
    # reproducible: collisions resolved by point index, not by scheduling
    again = npy(estimate_depth_kf_native(kf2, T(g["in_kld_gt"]), T(g["in_pose_gt"])))
    assert np.array_equal(gen, again)
    # mean=True (core/ops.py:84-92): scatter_reduce 'mean' with the zero-initialised image counted as one sample.  The far pose
    # piles several points onto every pixel it reaches; truncation flips at pixel borders move a point between neighbours, so
    # the bulk must agree tightly and the touched set almost exactly
    far = npy(estimate_depth_kf_native(kf2, T(g["in_kld_gt"]), T(g["in_pose_far"]), mean=True))
    assert ((far > 0) != (g["out_mean_far"] > 0)).mean() < 0.01
    assert np.isclose(far, g["out_mean_far"], rtol=1e-5).mean() > 0.97
    both = (far > 0) & (g["out_mean_far"] > 0)
    np.testing.assert_allclose(np.median(far[both] / g["out_mean_far"][both]), 1.0, rtol=1e-5)
    mg = npy(estimate_depth_kf_native(kf2, T(g["in_kld_gt"]), T(g["in_pose_gt"]), mean=True))
    assert np.isclose(mg, g["out_mean_general"], rtol=1e-5).mean() > 0.98
    assert np.array_equal(far, npy(estimate_depth_kf_native(kf2, T(g["in_kld_gt"]), T(g["in_pose_far"]), mean=True)))


def test_segment_reinit_and_average_match_reference():
    from super_primitive_amd.core import dense_optim
    from super_primitive_amd.depth_completion.segment_based_completion import average_visible_segments, render_depth_avg
    from super_primitive_amd.odometery.depth_init import segment_based_depth_reinit
    g = load_golden("g7_segment_stats")
    src, _ = frames_from_golden(g)
    for mode in ("mean", "median"):
        kld, seen = segment_based_depth_reinit(T(g["in_sparse_depth"]).clone(), src, mode=mode, return_info=True)
        assert np.array_equal(npy(seen), g[f"{mode}_visible"])
        np.testing.assert_allclose(npy(kld), g[f"{mode}_kld"], rtol=1e-5, atol=2e-6)
    kld, seen = T(g["median_kld"]), T(g["median_visible"])
    dense = dense_optim.unproject_kf_to_depths(src, kld)
    np.testing.assert_allclose(float(dense.double().sum()), float(g["depths_dense_sum"]), rtol=1e-6)
    np.testing.assert_allclose(npy(dense), g["depths_dense"], rtol=2e-3)       # fixture keeps an fp16 copy
    # fused table kernel
    avg, invalid = average_visible_segments(src, kld, seen)
    assert np.array_equal(npy(invalid), g["avg_invalid"])
    np.testing.assert_allclose(npy(avg), g["avg_depth"], rtol=2e-6, atol=1e-7)
    # dense-stack API form
    d = dense.clone()
    d[~src.keypoint_regions] = -1
    avg2, invalid2 = render_depth_avg(d[seen])
    assert np.array_equal(npy(invalid2), g["avg_invalid"])
    np.testing.assert_allclose(npy(avg2), g["avg_depth"], rtol=2e-6, atol=1e-7)


def test_kf_criteria_match_reference():
    """odometery/kf_criteria.py on the HIP path: the median is an element of the image (bit-exact), the translation
    ratio fp32, the angle within 1e-4 deg of scipy's (fp32 poses enter an fp64 atan2)."""
    from oracle import kf_oracle
    from super_primitive_amd.odometery.kf_criteria import keyframe_criterion, rotation_difference, translation_difference
    g = load_golden("g11_kf_criteria")
    for tag in ("odd", "even", "dense", "one"):
        a, b, d = T(g[f"{tag}_pose_src"]), T(g[f"{tag}_pose_trg"]), T(g[f"{tag}_depth"])
        diff, scale = translation_difference(a, b, d)
        assert diff.ndim == 0 and scale.ndim == 0 and diff.is_cuda
        assert float(scale) == float(g[f"{tag}_scale"])
        np.testing.assert_allclose(float(diff), float(g[f"{tag}_diff"]), rtol=2e-6)
        ang = rotation_difference(a, b)
        assert isinstance(ang, np.float64)
        np.testing.assert_allclose(ang, float(g[f"{tag}_angle_deg"]), rtol=2e-5, atol=1e-4)
        out = npy(keyframe_criterion(a, b, d))
        np.testing.assert_allclose(out[0], int(g[f"{tag}_n_valid"]) / d.numel(), rtol=1e-6)
    # full-size image with duplicates, holes and denormal-ish junk below the threshold, against the oracle
    rng = np.random.default_rng(9)
    depth = rng.uniform(0.2, 9.0, (480, 640)).astype(np.float32)
    depth[rng.uniform(size=depth.shape) < 0.4] = 0.0
    depth[rng.uniform(size=depth.shape) < 0.05] = 5e-7
    depth[100:200, 100:300] = 1.25                                        # a large plateau of equal values
    d = T(depth)
    a, b = T(g["odd_pose_src"]), T(g["odd_pose_trg"])
    want_diff, want_scale = kf_oracle.translation_difference(a.cpu(), b.cpu(), d.cpu())
    out = npy(keyframe_criterion(a, b, d))
    assert out[1] == float(want_scale)
    np.testing.assert_allclose(out[2], float(want_diff), rtol=2e-6)
    np.testing.assert_allclose(out[0], float(kf_oracle.validity_ratio(d.cpu())), rtol=1e-6)
    # nothing valid: the reference's torch.median raises on an empty tensor; the kernel reports NaN without syncing
    out = npy(keyframe_criterion(a, b, torch.zeros(64, 64, device=d.device)))
    assert out[0] == 0.0 and np.isnan(out[1]) and np.isnan(out[2])
    # the grid form (sp_kf_criterion_ws: what keyframe_criterion calls since round 6) against the one-workgroup form, bit for bit
    from super_primitive_amd import _lib
    lib = _lib.load()
    for img in (d, T(g["even_depth"]), T(g["one_depth"]), torch.zeros(64, 64, device=d.device)):
        one = torch.empty(4, dtype=torch.float32, device=d.device)
        _lib.check(lib.sp_kf_criterion(_lib.ptr(img.contiguous()), img.numel(), 1e-6, _lib.ptr(a.contiguous()), _lib.ptr(b.contiguous()), _lib.ptr(one), _lib.stream_ptr()), "sp_kf_criterion")
        np.testing.assert_array_equal(npy(one).view(np.uint32), npy(keyframe_criterion(a, b, img)).view(np.uint32))


def test_infer_depth_seeds_matches_oracle():
    from oracle import photometric_oracle as orc
    from super_primitive_amd import synth
    from super_primitive_amd.core import dense_optim
    from gpu_util import frames_from_synth
    pair = synth.make_pair(50, 70, 7, seed=21, shape="blobs")
    src, _ = frames_from_synth(pair)
    osrc, _ = orc.frames_from_synth(pair)
    got = dense_optim.infer_depth_seeds(T(pair.kld_init), src.keypoints, src.keypoint_regions, src.get_logdepth())
    want = orc.seed_logdepths(torch.from_numpy(pair.kld_init), osrc)
    np.testing.assert_allclose(npy(got), want.numpy(), rtol=0, atol=1e-6)


def test_lie_helpers_match_reference():
    from super_primitive_amd.lie import lie_algebra as la
    g = load_golden("g8_lie")
    out = la.renormalise_se3(T(g["in_noisy"]).clone())
    np.testing.assert_allclose(npy(out), g["renorm"], rtol=0, atol=1e-6)
    single = T(g["in_noisy"][2]).clone()
    assert la.renormalise_se3(single) is single                    # in place, like the reference
    np.testing.assert_allclose(npy(single), g["renorm"][2], rtol=0, atol=1e-6)
    np.testing.assert_allclose(npy(la.invertSE3(T(g["in_T"]))), g["inverse"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(npy(la.torch_pose_to_tq(T(g["in_T"]))), g["tq"], rtol=0, atol=1e-6)
    for i in range(4):
        np.testing.assert_allclose(npy(la.SE3_logmap(T(g["in_T"][i:i + 1])))[0], g["logmap"][i], rtol=1e-4, atol=1e-5)


def test_fused_se3_retraction_forward_and_backward():
    """sp_se3_retract vs the differentiable torch expression (itself checked against scipy expm on CPU)."""
    from super_primitive_amd.lie.se3 import SE3, LieGroupParameter, se3_exp_matrix
    rng = np.random.default_rng(3)
    for scale in (0.0, 1e-4, 3e-3, 0.05, 0.7, 2.5):
        n = 5
        a_np = (rng.standard_normal((n, 6)) * scale).astype(np.float32)
        X = SE3.exp(T((rng.standard_normal((n, 6)) * 0.5).astype(np.float32)))
        G = T(rng.standard_normal((n, 4, 4)).astype(np.float32))
        a1 = T(a_np, True)
        out1 = X.retr(a1).matrix()                       # fused
        (out1 * G).sum().backward()
        a2 = T(a_np, True).double()
        a2.retain_grad()
        out2 = se3_exp_matrix(a2) @ X.matrix().double()  # reference expression in fp64
        (out2 * G.double()).sum().backward()
        np.testing.assert_allclose(npy(out1), npy(out2), rtol=0, atol=3e-6)
        np.testing.assert_allclose(npy(a1.grad), npy(a2.grad), rtol=2e-5, atol=2e-5 * float(np.abs(npy(a2.grad)).max() + 1))
    p = LieGroupParameter(SE3.Identity(1, device="cuda:0"))
    m = p.retr().matrix()[0]
    m[:3, 3].sum().backward()
    np.testing.assert_allclose(npy(p.grad)[0, :3], np.ones(3), atol=1e-6)


def test_idempotence_properties():
    """renormalise_se3 is a projection; rebuilding the segment table gives identical arrays; packing is stable."""
    from super_primitive_amd import synth
    from super_primitive_amd.lie import lie_algebra as la
    from super_primitive_amd.segment_table import SegmentTable
    from gpu_util import frames_from_synth
    g = load_golden("g8_lie")
    once = la.renormalise_se3(T(g["in_noisy"]).clone())
    twice = la.renormalise_se3(once.clone())
    np.testing.assert_allclose(npy(twice), npy(once), atol=2e-7)
    pair = synth.make_pair(50, 70, 9, seed=4, shape="blobs")
    src, _ = frames_from_synth(pair)
    a = SegmentTable(src.keypoint_regions, src.logdepth_perseg, src.keypoints)
    b = SegmentTable(src.keypoint_regions, src.logdepth_perseg, src.keypoints)
    for f in ("pix", "baseL", "kp_L", "seg_off", "tiles", "seg_tile_off"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    # table order is torch.where order
    seg, row, col = torch.where(src.keypoint_regions)
    assert torch.equal((a.pix & 0xffff).long(), col) and torch.equal(((a.pix >> 16) & 0x7fff).long(), row)
    assert torch.equal(a.baseL, src.logdepth_perseg[seg, row, col])
    counts = torch.bincount(seg, minlength=9)
    assert torch.equal(a.seg_off[1:].long() - a.seg_off[:-1].long(), counts)


@pytest.mark.parametrize("H,W", [(480, 640), (45, 52), (33, 50), (2, 4)])
def test_fused_pyramid_step_and_packing_equals_the_two_passes(H, W):
    """Round 6 (set-up): ``sp_prepare_blur_pack`` -- one pyramid step of three-channel images together with the packed forms of its input and output
    levels -- against ``sp_prepare_blur`` followed by ``sp_prepare_pack``: bit for bit, on the 16-byte fast path (W a multiple of 4, even and odd
    H), on the general path, and with each optional output left out."""
    from super_primitive_amd import _lib
    from super_primitive_amd.optim.batch_prepare import stage
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(H * 1000 + W)
    imgs = [torch.rand(3, H, W, generator=g).to(dev) for _ in range(3)]
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    s = _lib.stream_ptr()
    # the two passes
    want_out = [torch.empty(3, Ho, Wo, device=dev) for _ in imgs]
    want_pin = [torch.empty(H, W, 3, device=dev) for _ in imgs]
    want_pout = [torch.empty(Ho, Wo, 3, device=dev) for _ in imgs]
    jb = np.zeros(len(imgs), dtype=np.dtype(_lib.SpPrepImage))
    jb['inp'], jb['out'], jb['H'], jb['W'] = [t.data_ptr() for t in imgs], [t.data_ptr() for t in want_out], H, W
    jp = np.zeros(2 * len(imgs), dtype=np.dtype(_lib.SpPrepImage))
    jp['inp'] = [t.data_ptr() for t in imgs] + [t.data_ptr() for t in want_out]
    jp['out'] = [t.data_ptr() for t in want_pin] + [t.data_ptr() for t in want_pout]
    jp['H'], jp['W'] = [H] * 3 + [Ho] * 3, [W] * 3 + [Wo] * 3
    st = stage([jb, jp], dev)
    _lib.check(lib.sp_prepare_blur(_lib.ptr(st[0]), 3, 3, Ho * Wo, s), "sp_prepare_blur")
    _lib.check(lib.sp_prepare_pack(_lib.ptr(st[1]), 6, H * W, s), "sp_prepare_pack")
    # the fused pass: job 0 everything, job 1 without the planar output, job 2 without the packed forms
    out = [torch.full((3, Ho, Wo), -1.0, device=dev) for _ in imgs]
    pin = [torch.full((H, W, 3), -1.0, device=dev) for _ in imgs]
    pout = [torch.full((Ho, Wo, 3), -1.0, device=dev) for _ in imgs]
    jf = np.zeros(len(imgs), dtype=np.dtype(_lib.SpPrepImagePack))
    jf['inp'], jf['H'], jf['W'] = [t.data_ptr() for t in imgs], H, W
    jf['out'] = [out[0].data_ptr(), 0, out[2].data_ptr()]
    jf['packed_in'] = [pin[0].data_ptr(), pin[1].data_ptr(), 0]
    jf['packed_out'] = [pout[0].data_ptr(), pout[1].data_ptr(), 0]
    sf = stage([jf], dev)
    _lib.check(lib.sp_prepare_blur_pack(_lib.ptr(sf[0]), 3, Ho * Wo, s), "sp_prepare_blur_pack")
    torch.cuda.synchronize()
    for k, (has_out, has_packed) in enumerate(((True, True), (False, True), (True, False))):
        assert torch.equal(out[k], want_out[k]) if has_out else bool((out[k] == -1.0).all())
        assert torch.equal(pin[k], want_pin[k]) if has_packed else bool((pin[k] == -1.0).all())
        assert torch.equal(pout[k], want_pout[k]) if has_packed else bool((pout[k] == -1.0).all())
