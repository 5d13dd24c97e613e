"""-m gpu: edge cases of the hot path -- degenerate visibility, extreme segment counts and sizes, cache
invalidation, argument limits.  Checked against the oracle where the reference semantics are defined."""
import numpy as np
import pytest
import torch

from gpu_util import T, frames_from_synth, npy

pytestmark = pytest.mark.gpu
CFG = {"mode": "colour", "collect_stats": 0}


def oracle_cost(pair, kld, pose):
    from oracle import photometric_oracle as orc
    src, trg = orc.frames_from_synth(pair)
    k = torch.from_numpy(kld).requires_grad_(True)
    p = torch.from_numpy(pose).requires_grad_(True)
    out = orc.photometric_cost(src, trg, k, p)
    out["residual"].sum().backward()
    return float(out["residual"]), k.grad.numpy(), p.grad.numpy()


def hip_cost(pair, kld, pose):
    from super_primitive_amd.core import dense_optim
    src, trg = frames_from_synth(pair)
    k, p = T(kld, True), T(pose, True)
    out = dense_optim.photomeric_cost(src, trg, k, p, CFG)
    out["residual"].sum().backward()
    return float(out["residual"]), npy(k.grad), npy(p.grad)


def test_nothing_visible_gives_exact_zero_and_finite_gradients():
    """Target camera looking the other way: every point invalid -> residual == 0 (mean over ALL points), zero grads."""
    from super_primitive_amd import synth
    pair = synth.make_pair(48, 64, 6, seed=3)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.diag([-1.0, 1.0, -1.0]).astype(np.float32)      # 180 degrees about y: all z' < 0
    r, gk, gp = hip_cost(pair, pair.kld_init, pose)
    r0, gk0, gp0 = oracle_cost(pair, pair.kld_init, pose)
    assert r == 0.0 and r0 == 0.0
    assert np.all(gk == 0) and np.all(gp == 0) and np.all(np.isfinite(gp))


def test_single_segment_covering_the_frame_and_many_tiny_segments():
    from super_primitive_amd import synth
    one = synth.make_pair(64, 96, 1, seed=11)
    r, gk, gp = hip_cost(one, one.kld_init, one.pose_init)
    r0, gk0, gp0 = oracle_cost(one, one.kld_init, one.pose_init)
    np.testing.assert_allclose(r, r0, rtol=2e-5)
    np.testing.assert_allclose(gk, gk0, rtol=2e-3, atol=1e-7)
    assert np.abs(gp - gp0).max() <= 2e-3 * np.abs(gp0).max()
    # 330 segments: 1-pixel ones, empty ones, overlapping ones (VOID-like counts)
    many = synth.make_pair(60, 88, 330, seed=12, shape="blobs")
    m = many.keypoint_regions
    for n in range(0, 330, 7):
        m[n] = False
        m[n, many.meta["kp_rc"][n, 0], many.meta["kp_rc"][n, 1]] = True      # single pixel
    for n in range(3, 330, 50):
        m[n] = False                                                           # empty
    many.logdepth_perseg[~m] = 0
    r, gk, gp = hip_cost(many, many.kld_init, many.pose_init)
    r0, gk0, gp0 = oracle_cost(many, many.kld_init, many.pose_init)
    np.testing.assert_allclose(r, r0, rtol=2e-5)
    assert np.abs(gk - gk0).max() <= 2e-3 * np.abs(gk0).max()
    assert np.all(gk[3::50] == 0), "empty segments receive no gradient"
    assert np.abs(gp - gp0).max() <= 2e-3 * np.abs(gp0).max()


def test_large_frame_single_pass():
    """1080 x 1440, 12 segments (P ~ 1.6 M points): exercises multi-tile segments and 32-bit offsets."""
    from super_primitive_amd import synth
    pair = synth.make_pair(1080, 1440, 12, seed=13, init_sigma=0.003)
    r, gk, gp = hip_cost(pair, pair.kld_init, pair.pose_init)
    r0, gk0, gp0 = oracle_cost(pair, pair.kld_init, pair.pose_init)
    np.testing.assert_allclose(r, r0, rtol=2e-5)
    assert np.abs(gk - gk0).max() <= 2e-3 * np.abs(gk0).max()
    assert np.abs(gp - gp0).max() <= 2e-3 * np.abs(gp0).max()


def test_table_cache_follows_tensor_replacement_and_inplace_edits():
    """Drivers replace keypoint_regions / logdepth_perseg wholesale (frontend/segment/post_processer.py:176-179);
    the compact table must be rebuilt, also after an in-place edit (version counter)."""
    from super_primitive_amd import synth
    from super_primitive_amd.core import dense_optim
    from super_primitive_amd.segment_table import table_of
    pair = synth.make_pair(48, 64, 6, seed=14)
    src, trg = frames_from_synth(pair)
    kld, pose = T(pair.kld_init), T(pair.pose_init)
    r1 = float(dense_optim.photomeric_cost(src, trg, kld, pose, CFG)["residual"])
    t1 = table_of(src)
    assert table_of(src) is t1
    new_masks = src.keypoint_regions.clone()
    new_masks[0, :10] = False
    src.keypoint_regions = new_masks
    t2 = table_of(src)
    assert t2 is not t1 and t2.P < t1.P
    r2 = float(dense_optim.photomeric_cost(src, trg, kld, pose, CFG)["residual"])
    pair.keypoint_regions[0, :10] = False
    np.testing.assert_allclose(r2, oracle_cost(pair, pair.kld_init, pair.pose_init)[0], rtol=2e-5)
    assert r2 != r1
    src.logdepth_perseg[1] += 0.1 * src.keypoint_regions[1]           # in place: bumps ._version
    assert table_of(src) is not t2
    pair.logdepth_perseg[1] += np.float32(0.1) * pair.keypoint_regions[1]
    r3 = float(dense_optim.photomeric_cost(src, trg, kld, pose, CFG)["residual"])
    np.testing.assert_allclose(r3, oracle_cost(pair, pair.kld_init, pair.pose_init)[0], rtol=2e-5)


def test_tiny_pyramid_level_and_argument_limits():
    from super_primitive_amd import _lib, synth
    from super_primitive_amd.core import dense_optim
    from super_primitive_amd.image.keyframe import keyframe_pyramid
    pair = synth.make_pair(16, 24, 2, seed=15)
    src, trg = frames_from_synth(pair)
    sp, tp = keyframe_pyramid(src, 0, 4), keyframe_pyramid(trg, 0, 4)         # coarsest level is 2 x 3 pixels
    assert tuple(sp[0].image.shape[-2:]) == (2, 3)
    out = dense_optim.photomeric_cost(sp[0], tp[0], T(pair.kld_init), T(pair.pose_init), CFG)
    assert np.isfinite(npy(out["residual"])).all()
    from oracle import photometric_oracle as orc
    osrc, otrg = orc.frames_from_synth(pair)
    want = orc.photometric_cost(orc.frame_pyramid(osrc, 0, 4)[0], orc.frame_pyramid(otrg, 0, 4)[0],
                                torch.from_numpy(pair.kld_init), torch.from_numpy(pair.pose_init))["residual"]
    np.testing.assert_allclose(npy(out["residual"]), want.numpy(), rtol=5e-5)
    lib = _lib.load()
    assert lib.sp_mask_count(_lib.ptr(src.keypoint_regions), 2, 40000, 24, _lib.ptr(src.keypoint_regions),
                             _lib.ptr(src.keypoint_regions), _lib.ptr(src.keypoint_regions), None) == -2   # SP_ELIMIT


def test_schedule_with_an_empty_decimated_point_set():
    """A lattice stride larger than every segment leaves the coarse table EMPTY (no mask pixel on the lattice): the scheduled
    run must pass through that phase (no launch for it, no hang, no division by zero) and still converge on the full points."""
    from super_primitive_amd import synth
    from super_primitive_amd.optim.pair_batch import PairBatch
    pr = synth.make_pair(24, 32, 4, seed=121, init_sigma=0.002)
    pr.keypoint_regions[:, ::2, :] = False                      # no pixel with an even row survives ...
    pr.keypoint_regions[:, :, ::2] = False                      # ... nor with an even column: strides 2 and 4 find nothing
    batch = PairBatch.from_synth([pr], levels=(0, 3), device="cuda:0", tile_points=512, point_stride=(1, 2, 4))
    assert all(lay.n_spans == 0 and lay.points == [0] for lay in batch.coarse.values())
    launched = batch.run_scheduled(max_iters_per_level=10, conv_tol=1e-3, polish_max=5, polish_eps=1e-5, polish_tol=1e-4)
    torch.cuda.synchronize()
    assert 0 < launched <= 35 and int(batch.phase[0]) == 4
    assert np.isfinite(batch.poses().cpu().numpy()).all() and np.isfinite(batch.klds()[0].cpu().numpy()).all()
    ref = PairBatch.from_synth([pr], levels=(0, 3), device="cuda:0", tile_points=512)
    c0 = float(ref.evaluate(0)[0])
    assert float(batch.evaluate(0)[0]) < c0
