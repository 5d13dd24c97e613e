"""CPU-only: host-side logic (work-list construction, synthetic scenes, SE(3) glue, API surface) and the C ABI
library: it loads and exports every symbol include/sp_hip.h declares.  No compute kernel is called."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from super_primitive_amd import _lib
    header = open(os.path.join(ROOT, "include", "sp_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\bint\s+(sp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sp_abi_version() == _lib.SP_ABI_VERSION
    # argument counts of the ctypes table match the prototypes
    for name in declared:
        proto = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", header, flags=re.S).group(1).strip()
        n = 0 if proto in ("void", "") else proto.count(",") + 1
        assert n == len(_lib.SIGNATURES[name]), name


def test_abi_constants_and_struct_layout_match_header():
    from super_primitive_amd import _lib
    header = open(os.path.join(ROOT, "include", "sp_hip.h")).read()
    for macro in ("SP_ABI_VERSION", "SP_GRAD_PARTIAL_FLOATS", "SP_GN_PARTIAL_FLOATS", "SP_LM_STATE_FLOATS"):
        val = int(re.search(r"#define\s+" + macro + r"\s+(\d+)", header).group(1))
        assert getattr(_lib, macro) == val, macro
    assert ctypes.sizeof(_lib.SpPair) == 136
    assert _lib.SpPair.K_src.offset == 64 and _lib.SpPair.N.offset == 96 and _lib.SpPair.zmin.offset == 128
    # window optimiser structs (sizes are static_assert-ed on the device side, sp_window.hip)
    assert ctypes.sizeof(_lib.SpWindowNode) == 176 and ctypes.sizeof(_lib.SpWindowEdge) == 16 and ctypes.sizeof(_lib.SpWindowBlock) == 32
    assert _lib.SpWindowNode.a.offset == 64 and _lib.SpWindowNode.aff.offset == 136 and _lib.SpWindowNode.lr_pose.offset == 160
    assert _lib.SpWindowNode.kind.offset == 168 and _lib.SpWindowBlock.N.offset == 24
    # per-pair schedule, passed by value
    assert int(re.search(r"#define\s+SP_MAX_PHASES\s+(\d+)", header).group(1)) == _lib.SP_MAX_PHASES
    assert ctypes.sizeof(_lib.SpPrepTable) == 240 and _lib.SpPrepTable.bits.offset == 224 and _lib.SpPrepTable.boxes.offset == 232
    assert ctypes.sizeof(_lib.SpPrepTable) == 240 and ctypes.sizeof(_lib.SpPrepSample) == 176 and ctypes.sizeof(_lib.SpPrepImage) == 24
    assert _lib.SpPrepTable.stride.offset == 192 and _lib.SpPrepSample.N.offset == 152 and _lib.SpPrepImage.H.offset == 16
    for macro in ("SP_PREP_MAX_STRIDES", "SP_PREP_MAX_LEVELS"):
        assert int(re.search(r"#define\s+" + macro + r"\s+(\d+)", header).group(1)) == getattr(_lib, macro)
    assert ctypes.sizeof(_lib.SpPhase) == 64 and _lib.SpPhase.n_spans.offset == 40 and _lib.SpPhase.conv_tol.offset == 52
    assert _lib.SpPhase.flags.offset == 56
    assert _lib.SpPhase.next.offset == 60
    # (ABI 13: 12 phases, the third attempt's entry, the Adam phases' rates and moments; the within-pair thresholds of the verdict)
    assert ctypes.sizeof(_lib.SpSchedule) == 800 and _lib.SpSchedule.n_phases.offset == 768 and _lib.SpSchedule.retry_entry.offset == 776
    assert _lib.SpSchedule.retry2_entry.offset == 780 and _lib.SpSchedule.adam_lr_pose.offset == 784 and _lib.SpSchedule.adam_state.offset == 792
    assert ctypes.sizeof(_lib.SpVerdict) == 104 and _lib.SpVerdict.evals.offset == 88 and _lib.SpVerdict.seg_product.offset == 96 and _lib.SpVerdict.kld_bound.offset == 56 and _lib.SpVerdict.lam0.offset == 76
    assert _lib.SpVerdict.seg_max_ratio.offset == 80 and _lib.SpVerdict.seg_mean_ratio.offset == 84
    assert ctypes.sizeof(_lib.SpQueue) == 296 and _lib.SpQueue.active.offset == 288 and _lib.SpQueue.max_spans.offset == 192 and _lib.SpQueue.head.offset == 248
    for macro in ("SP_GN_SEG_FLOATS", "SP_GNA_SEG_FLOATS", "SP_PHASE_ADAM", "SP_VERDICT_SEGMENTS", "SP_VERDICT_SEGMENT_POINTS", "SP_VERDICT_MIN_SEGMENTS"):
        assert int(re.search(r"#define\s+" + macro + r"\s+(\d+)", header).group(1)) == getattr(_lib, macro), macro
    for macro in ("SP_STATUS_NONFINITE", "SP_STATUS_LAST_CAP", "SP_STATUS_DEPTH_RANGE", "SP_STATUS_COST", "SP_STATUS_VALID", "SP_STATUS_SEGMENTS",
                  "SP_STATUS_RETRIED", "SP_STATUS_ADAM", "SP_STATUS_UNFINISHED", "SP_DIAG_FLOATS"):
        assert int(re.search(r"#define\s+" + macro + r"\s+(0x[0-9a-fA-F]+|\d+)", header).group(1), 0) == getattr(_lib, macro), macro
    assert int(re.search(r"#define\s+SP_PHASE_POSE_ONLY\s+(\d+)", header).group(1)) == _lib.SP_PHASE_POSE_ONLY
    # (ABI 14: the argument record of sp_chain_step; static_assert-ed in sp_chain.hip)
    assert ctypes.sizeof(_lib.SpChainPhase) == 16 and ctypes.sizeof(_lib.SpChainWindow) == 680 and ctypes.sizeof(_lib.SpChainTarget) == 56
    assert ctypes.sizeof(_lib.SpChainStep) == 1752 and _lib.SpChainStep.track.offset == 56 and _lib.SpChainStep.supp.offset == 808
    assert _lib.SpChainWindow.state_host.offset == 672 and _lib.SpChainWindow.check_first.offset == 668 and _lib.SpChainStep.crit_ws.offset == 1720
    for macro in ("SP_CHAIN_LEVELS", "SP_CHAIN_PHASES", "SP_CHAIN_TRACK", "SP_CHAIN_SUPP", "SP_CHAIN_CRITERION"):
        assert int(re.search(r"#define\s+" + macro + r"\s+(\d+)", header).group(1)) == getattr(_lib, macro), macro
    lib = _lib.load()
    assert lib.sp_chain_step(None, None) == -1 and lib.sp_kf_criterion_ws_words() == 4 * 256 + 8
    step = _lib.SpChainStep()
    assert lib.sp_chain_step(ctypes.byref(step), None) == -1               # (no frame size, no stage's pointers: refused before any launch)


def test_new_entry_points_validate_arguments_and_sizes():
    from super_primitive_amd import _lib
    lib = _lib.load()
    assert lib.sp_points_workspace_floats(5000, 2) == 3 * 2 * 16 and lib.sp_points_workspace_floats(0, 1) == 0
    assert lib.sp_points_cost_grad(*([None] * 2), 10, 10, 4, 4, None, 2, 2, None, None, 1, None, None, 1e-7, *([None] * 5)) == -1
    assert lib.sp_window_scratch_doubles(10, 40) == 10 * (28 + 40)
    assert lib.sp_window_compose(None, None, 1, None, 1, None) == -1
    assert lib.sp_window_step(*([None] * 2), 1, None, 1, None, 1, 1, *([None] * 3), 0, 0, 0.0, None, None, 0, None) == -1
    for fn in (lib.sp_prepare_count, lib.sp_prepare_fill):
        assert fn(None, 1, 1, 1, None) == -1
    assert lib.sp_prepare_sample(None, 1, 1, None) == -1 and lib.sp_prepare_pack(None, 1, 1, None) == -1
    assert lib.sp_prepare_blur(None, 1, 3, 1, None) == -1
    sched = _lib.SpSchedule()
    assert lib.sp_pairs_schedule_cost(ctypes.addressof(sched), None, None) == -1 and lib.sp_pairs_schedule_cost(None, None, None) == -1
    assert lib.sp_pairs_schedule_gn_step(ctypes.addressof(sched), 1, 1, 8.0, 0.5, 1e-7, *([None] * 7)) == -1
    assert lib.sp_depth_accumulate(*([None] * 6), 1, 1, 1, 1, None, None) == -1
    assert lib.sp_depth_average_finish(None, 4, 4, None, None, None) == -1


def test_work_list_helpers_cover_every_point_once():
    """pad_layout / build_work_list (shared by PairBatch and PoseWindow): padded runs are granule multiples, chunks tile every
    padded segment exactly once, spans tile the chunks, record offsets count 4 records per chunk."""
    from super_primitive_amd.optim.pair_batch import GRANULE, build_work_list, pad_layout
    rng = np.random.default_rng(3)
    pads = [pad_layout(rng.integers(0, 3000, size=n), "cpu") for n in (7, 1, 30)]
    wl = build_work_list(pads, span_points=2048, tile_points=1024)
    chunks, spans = wl["chunks"], wl["spans"]
    assert np.all(chunks[:, 3] % GRANULE == 0) and np.all(chunks[:, 3] > 0) and np.all(chunks[:, 3] <= 1024)
    for m, pd in enumerate(pads):
        mine = chunks[chunks[:, 0] == m]
        covered = np.zeros(pd["Ppad"], dtype=int)
        for _, seg, start, count in mine:
            assert pd["pseg_off"][seg] <= start and start + count <= pd["pseg_off"][seg] + pd["pc"][seg]
            covered[start:start + count] += 1
        assert np.all(covered == 1)
        assert wl["seg_rec_offs"][m][-1] == 4 * len(mine)
    assert spans[:, 1].sum() == len(chunks) and np.array_equal(np.cumsum(spans[:, 1]) - spans[:, 1], spans[:, 0])
    for q, n, pts, pair in spans:
        assert pts == chunks[q:q + n, 3].sum() and np.all(chunks[q:q + n, 0] == pair)


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any HIP call, so it can be exercised here."""
    from super_primitive_amd import _lib
    lib = _lib.load()
    assert lib.sp_pairs_cost(None, None, None, 0, 0, 0.0, None, None, None) == -1
    assert lib.sp_blur_decimate(None, 3, 10, 10, None, None) == -1
    assert lib.sp_renormalise_se3(None, 1, None) == -1
    with pytest.raises(RuntimeError, match="SP_EINVAL"):
        _lib.check(-1, "x")


def test_cpu_tensors_are_refused_by_the_product_path():
    from super_primitive_amd import synth
    from super_primitive_amd.core import dense_optim, dense_optim_batch
    from super_primitive_amd.core.depth_render import estimate_depth_kf_native
    from super_primitive_amd.image.keyframe import KeyFrame
    pair = synth.make_pair(24, 32, 2, seed=1)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    src = KeyFrame(c(pair.src_image), c(pair.K), c(pair.logdepth_perseg), c(pair.keypoints), c(pair.keypoint_regions))
    trg = KeyFrame(c(pair.trg_image), c(pair.K))
    cfg = {"mode": "colour", "collect_stats": 0}
    with pytest.raises(RuntimeError, match="HIP-only"):
        dense_optim.photomeric_cost(src, trg, c(pair.kld_init), c(pair.pose_init), cfg)
    with pytest.raises(RuntimeError, match="HIP-only"):
        dense_optim_batch.photomeric_cost_batch(src, c(pair.trg_image)[None], c(pair.K)[None], c(pair.kld_init),
                                                c(pair.pose_init)[None], cfg)
    with pytest.raises(RuntimeError, match="HIP-only"):
        dense_optim.unproject_kf(src, c(pair.kld_init))
    with pytest.raises(RuntimeError, match="HIP-only"):
        estimate_depth_kf_native(src, c(pair.kld_init))
    with pytest.raises(NotImplementedError):
        dense_optim.photomeric_cost(src, trg, c(pair.kld_init), c(pair.pose_init), {"mode": "colour_norm", "collect_stats": 0})


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under super_primitive_amd/ may import it."""
    pkg = os.path.join(ROOT, "super_primitive_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


@pytest.mark.parametrize("counts,tp", [([5829, 0, 100, 4096, 4097, 513], 4096), ([1], 256), ([300000, 7], 8192), ([0, 0, 5], 1024)])
def test_make_tiles_partitions_every_segment(counts, tp):
    from super_primitive_amd.segment_table import make_tiles
    tiles, off = make_tiles(counts, tp, pair=3, first_point=10)
    assert off[0] == 0 and off[-1] == len(tiles) and len(off) == len(counts) + 1
    pos = 10
    for n, c in enumerate(counts):
        mine = tiles[off[n]:off[n + 1]]
        assert (mine[:, 0] == 3).all() and (mine[:, 1] == n).all()
        assert (mine[:, 3] > 0).all() and (mine[:, 3] <= tp).all()
        assert mine[:, 3].sum() == c
        if len(mine):
            assert mine[0, 2] == pos and np.array_equal(mine[1:, 2], mine[:-1, 2] + mine[:-1, 3])
            assert (mine[:-1, 3] % 256 == 0).all()
            assert mine[:, 3].max() - mine[:, 3].min() <= 256 + (mine[:, 3].max() - mine[-1, 3])
        pos += c


def test_synth_pair_is_deterministic_and_consistent():
    from super_primitive_amd import synth
    a, b = synth.make_pair(40, 56, 6, seed=5, shape="blobs"), synth.make_pair(40, 56, 6, seed=5, shape="blobs")
    for f in ("src_image", "trg_image", "logdepth_perseg", "keypoints", "kld_gt", "pose_gt", "pose_init"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.keypoint_regions.dtype == bool and a.keypoint_regions.shape == (6, 40, 56)
    assert np.all(a.logdepth_perseg[~a.keypoint_regions] == 0)
    rc = np.round(0.5 * (np.array([40, 56]) - 1) * (a.keypoints + 1)).astype(int)
    for n in range(6):
        assert a.keypoint_regions[n, rc[n, 0], rc[n, 1]]
    R = a.pose_gt[:3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
    assert 0 <= a.src_image.min() and a.src_image.max() <= 1


def test_se3_parameter_matches_matrix_exponential_and_autograd():
    from scipy.linalg import expm
    from super_primitive_amd.lie.se3 import SE3, LieGroupParameter, se3_exp_matrix
    rng = np.random.default_rng(0)
    for scale in (0.0, 1e-5, 0.2, 1.5):
        xi = rng.standard_normal(6) * scale
        M = np.zeros((4, 4))
        M[:3, :3] = [[0, -xi[5], xi[4]], [xi[5], 0, -xi[3]], [-xi[4], xi[3], 0]]
        M[:3, 3] = xi[:3]
        np.testing.assert_allclose(se3_exp_matrix(torch.from_numpy(xi)[None])[0].numpy(), expm(M), rtol=1e-9, atol=1e-12)
    X = SE3.exp(torch.tensor([[0.1, -0.2, 0.3, 0.2, 0.1, -0.3]]))
    p = LieGroupParameter(X)
    assert p.is_leaf and p.requires_grad and tuple(p.shape) == (1, 6) and float(p.abs().sum()) == 0
    T = p.retr().matrix()[0]
    np.testing.assert_allclose(T.detach().numpy(), X.matrix()[0].numpy(), atol=1e-7)      # Exp(0) * X
    T[:3, 3].sum().backward()
    assert torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    # left retraction: d t / d tau = I at a = 0
    np.testing.assert_allclose(p.grad[0, :3].numpy(), np.ones(3), atol=1e-6)
    # group round trips
    Y = SE3.InitFromVec(X.vec())
    np.testing.assert_allclose(Y.matrix().numpy(), X.matrix().numpy(), atol=1e-6)
    np.testing.assert_allclose(X.mul(X.inv()).matrix()[0].numpy(), np.eye(4), atol=1e-6)


def test_reference_module_surface_is_present():
    """Names the reference's drivers import."""
    import super_primitive_amd.core.dense_optim as do
    import super_primitive_amd.core.dense_optim_batch as dob
    import super_primitive_amd.core.ops as ops
    import super_primitive_amd.core.depth_render as dr
    import super_primitive_amd.image.gaussian_pyramid as gp
    import super_primitive_amd.image.keyframe as kf
    import super_primitive_amd.lie.lie_algebra as la
    import super_primitive_amd.lie.lietorch_utils as lu
    import super_primitive_amd.tool.point_utils as pu
    import super_primitive_amd.odometery.depth_init as di
    for mod, names in [
        (do, "photomeric_cost photomeric_cost_precomputed unproject_kf unproject_kf_to_depths infer_depth_seeds expdepth "
             "unproject_segments unproject_points transform_points img_interp get_pixels affine_compensation_batch_v2 "
             "calculate_residual project_points infer_spatial_size"),
        (dob, "photomeric_cost_batch get_pixels_batch"),
        (ops, "transform_points_batch project_points_batch project_points transform_points unproject_points_mat estimate_depth_diff"),
        (dr, "estimate_depth_kf_native"),
        (gp, "GaussianBlurModule ImagePyramidModule DepthPyramidModule IntrinsicsPyramidModule pyr_depth resize_intrinsics resize_depth"),
        (kf, "KeyFrame keyframe_pyramid put_keypoints_back put_keypoints_back_kf infer_spatial_size"),
        (la, "quaternion_to_matrix renormalise_se3 torch_pose_to_tq matrix_to_q_torch pose_to_tq tq_to_pose se3_exp batch_se3 "
             "invertSE3 normalizeSE3_inplace SO3_expmap SO3_logmap skew_symmetric SE3_logmap"),
        (lu, "lietorch_detach lietorch_new_param zero_out_lietorch_tensor mat_to_lie print_pose"),
        (pu, "normalise_coordinates denormalise_coordinates normalise_coordinates_np denormalise_coordinates_np"),
        (di, "segment_based_depth_reinit"),
    ]:
        for n in names.split():
            assert hasattr(mod, n), f"{mod.__name__}.{n}"


def test_install_as_reference_modules_aliases_the_packages():
    import importlib
    import sys
    import super_primitive_amd
    saved = {k: sys.modules.get(k) for k in ("core", "core.dense_optim", "image", "lie", "tool", "odometery", "depth_completion", "frontend",
                                             "frontend.segment", "frontend.segment.post_processer")}
    try:
        super_primitive_amd.install_as_reference_modules()
        import core.dense_optim as do
        from super_primitive_amd.core import dense_optim
        assert do is dense_optim
        assert importlib.import_module("image.keyframe").KeyFrame is super_primitive_amd.image.keyframe.KeyFrame
        # the keyframe post-processing (N2) under the reference's import path (frontend/process_frame.py: ``from frontend.segment import post_processer``)
        import frontend.segment.post_processer as pp
        from super_primitive_amd.frontend.segment import post_processer
        assert pp is post_processer and hasattr(pp, "kf_fix_disconnected_regions")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in list(sys.modules):
            if k.split(".")[0] in ("core", "image", "lie", "tool", "odometery", "depth_completion", "frontend") and k not in saved:
                sys.modules.pop(k, None)


def test_lazy_stats_dict_semantics():
    """LazyStats behaves like the plain dict it stands for; the producer runs once, on first look."""
    import pickle
    from super_primitive_amd.core.dense_optim import LazyStats
    calls = []

    def producer():
        calls.append(1)
        return {"src_pts": torch.arange(6.).reshape(2, 3), "median_depth": None}

    d = LazyStats(torch.tensor([0.25]), producer)
    assert float(d["residual"]) == 0.25 and d.get("residual") is not None and "residual" in d and not calls
    assert "src_pts" in d and calls == [1]
    assert set(d.keys()) == {"residual", "src_pts", "median_depth"} and len(d) == 3 and d["median_depth"] is None
    assert {k for k, _ in d.items()} == set(d) and len(list(d.values())) == 3 and calls == [1]
    e = LazyStats(torch.tensor([0.5]), producer)
    back = pickle.loads(pickle.dumps(e))
    assert type(back) is dict and set(back) == {"residual", "src_pts", "median_depth"} and calls == [1, 1]
    assert type(LazyStats(torch.tensor([1.0]), producer).copy()) is dict


def test_thin_helpers_match_the_reference():
    """Mirrors that are plain tensor expressions (depth pyramid steps, pyramid level selection, normal-channel rotation,
    host conversions) against outputs of the real reference modules (golden g13)."""
    import numpy as np
    from conftest import load_golden
    from super_primitive_amd.core import normal_cost
    from super_primitive_amd.image import gaussian_pyramid as gp
    from super_primitive_amd.tool import etc
    g = load_golden("g13_helpers")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for mode in ("bilinear", "nearest_neighbor", "max", "min"):
        np.testing.assert_array_equal(gp.pyr_depth(T(g["depth"]), mode, 2).numpy(), g[f"pyr_{mode}"])
    np.testing.assert_allclose(gp.pyr_depth(T(g["holes"]), "masked_bilinear", 2).numpy(), g["pyr_masked_bilinear"], rtol=1e-6)
    with pytest.raises(ValueError):
        gp.pyr_depth(T(g["depth"]), "cubic", 2)
    big = T(np.tile(g["depth"], (1, 1, 4, 4)))
    K = T(np.array([[500., 0, 320], [0, 510., 240], [0, 0, 1]], np.float32))
    for (s0, e0) in ((0, 3), (1, 4), (2, 3), (0, 1)):
        lv = gp.DepthPyramidModule(s0, e0, "nearest_neighbor", "cpu")(big)
        assert len(lv) == int(g[f"dpyr_{s0}_{e0}_n"])
        for i, l in enumerate(lv):
            np.testing.assert_array_equal(l.numpy(), g[f"dpyr_{s0}_{e0}_{i}"])
        Ks = torch.stack(gp.IntrinsicsPyramidModule(s0, e0, "cpu")(K, [1.0, 0.5]))
        np.testing.assert_allclose(Ks.numpy(), g[f"kpyr_{s0}_{e0}"], rtol=1e-6)
    for mode, C in (("colour", 3), ("colour_norm", 6), ("colour_norm_kappa", 7)):
        got = normal_cost.transform_normals_batch(T(g["px"][:, :C]), T(g["poses"]), mode)
        np.testing.assert_allclose(got.numpy(), g[f"nrm_{mode}"], rtol=1e-6, atol=1e-7)
        got = normal_cost.transform_normals(T(g["px"][:, :C]), T(g["poses"][1]), mode)
        np.testing.assert_allclose(got.numpy(), g[f"nrm1_{mode}"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(etc.to_img(T(g["img"])), g["to_img"])
    np.testing.assert_array_equal(etc.to_img_np(T(g["img"])), g["to_img_np"])
    np.testing.assert_array_equal(etc.image_tt(g["u8"], device="cpu").numpy(), g["image_tt"])
    arr = g["img"]
    assert etc.to_np(arr) is arr and torch.equal(etc.from_np(arr), T(arr)) and etc.from_np(arr).data_ptr() != arr.ctypes.data
    d = etc.dict_cpu({"a": T(g["img"]), "_sp": object(), "n": None})
    assert set(d) == {"a", "n"} and d["n"] is None


def test_vectorised_layout_and_work_list_equal_the_per_pair_ones():
    """batch_prepare.flat_layout / flat_work_list (numpy over the whole batch, used by PairBatch) produce exactly the chunks,
    spans and record offsets of pad_layout / build_work_list (per pair, per segment), including empty segments, segments
    longer than a chunk, and pairs with different segment counts."""
    from super_primitive_amd.optim import batch_prepare
    from super_primitive_amd.optim.pair_batch import build_work_list, pad_layout
    rng = np.random.default_rng(11)
    for tile_points, span_points in ((1024, 2048), (8192, 16384), (256, 256), (2048, 100000)):
        Ns = [7, 1, 30, 12]
        counts = [rng.integers(0, 6000, size=n) for n in Ns]
        counts[2][3] = 0
        counts[0][0] = 256
        pads = [pad_layout(c, "cpu") for c in counts]
        ref = build_work_list(pads, span_points, tile_points)
        n_off = np.concatenate(([0], np.cumsum(Ns)))
        pc, seg_pos, p_off = batch_prepare.flat_layout(np.concatenate(counts), n_off)
        assert np.array_equal(pc, np.concatenate([pd['pc'] for pd in pads]))
        assert np.array_equal(seg_pos, np.concatenate([pd['pseg_off'][:-1] for pd in pads]))
        assert np.array_equal(np.diff(p_off), [pd['Ppad'] for pd in pads])
        for build in (batch_prepare.flat_work_list, batch_prepare.flat_work_list_numpy):      # the library's host helper and its numpy statement
            got = build(pc, seg_pos, n_off, span_points, tile_points)
            assert np.array_equal(got['chunks'], ref['chunks']) and np.array_equal(got['spans'], ref['spans'])
            assert np.array_equal(got['seg_tile_off'], np.concatenate(ref['seg_rec_offs']))
            assert np.array_equal(got['c_off'], ref['c_off']) and np.array_equal(got['s_off'], ref['s_off'])
            assert np.array_equal(got['sto_off'], np.concatenate(([0], np.cumsum([n + 1 for n in Ns]))))
    # nothing at all
    got = batch_prepare.flat_work_list(np.zeros(3, np.int64), np.zeros(3, np.int64), np.array([0, 3]), 1024, 1024)
    assert got['chunks'].shape == (0, 4) and got['spans'].shape == (0, 4) and np.array_equal(got['seg_tile_off'], [0, 0, 0, 0])


def test_vectorised_work_list_equals_the_loop_on_random_layouts():
    """Property test (hypothesis): for random segment counts (empty segments, segments longer than a chunk, single-segment
    pairs), chunk and span sizes, flat_layout / flat_work_list reproduce pad_layout / build_work_list exactly."""
    from hypothesis import given, settings, strategies as st
    from super_primitive_amd.optim import batch_prepare
    from super_primitive_amd.optim.pair_batch import build_work_list, pad_layout

    pair = st.lists(st.one_of(st.just(0), st.integers(1, 9000)), min_size=1, max_size=12)

    @settings(max_examples=60, deadline=None)
    @given(st.lists(pair, min_size=1, max_size=5), st.sampled_from([256, 512, 1000, 2048, 8192]), st.sampled_from([256, 700, 4096, 16384, 10 ** 6]))
    def check(counts, tile_points, span_points):
        pads = [pad_layout(np.asarray(c), "cpu") for c in counts]
        ref = build_work_list(pads, span_points, tile_points)
        n_off = np.concatenate(([0], np.cumsum([len(c) for c in counts])))
        pc, seg_pos, p_off = batch_prepare.flat_layout(np.concatenate([np.asarray(c) for c in counts]), n_off)
        for build in (batch_prepare.flat_work_list, batch_prepare.flat_work_list_numpy):
            got = build(pc, seg_pos, n_off, span_points, tile_points)
            assert np.array_equal(got['chunks'].reshape(-1, 4), ref['chunks']) and np.array_equal(got['spans'].reshape(-1, 4), ref['spans'])
            assert np.array_equal(got['seg_tile_off'], np.concatenate(ref['seg_rec_offs']))
            assert np.array_equal(got['c_off'], ref['c_off']) and np.array_equal(got['s_off'], ref['s_off'])
        assert np.array_equal(np.diff(p_off), [pd['Ppad'] for pd in pads])

    check()


def test_host_layout_helper_equals_the_numpy_layout():
    """sp_host_layout (one pass in C over all lattices) against optim.batch_prepare.flat_layout (the numpy statement), ragged pairs, empty
    segments, both granules."""
    import ctypes
    import numpy as np
    from super_primitive_amd import _lib
    from super_primitive_amd.optim.batch_prepare import flat_layout
    lib = _lib.load()
    rng = np.random.default_rng(5)
    Ns = np.array([6, 1, 9, 4, 64], dtype=np.int64)
    n_off = np.concatenate(([0], np.cumsum(Ns)))
    S, M, nL = int(n_off[-1]), len(Ns), 3
    counts = rng.integers(0, 2000, size=(nL, S)).astype(np.int32)
    counts[1, ::5] = 0
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for granule in (256, 64):
        pc, seg_pos = np.empty((nL, S), np.int64), np.empty((nL, S), np.int64)
        p_off, points = np.empty((nL, M + 1), np.int64), np.empty((nL, M), np.int64)
        seg_off = np.empty((nL, 2 * S), np.int32)
        assert lib.sp_host_layout(vp(counts), nL, S, vp(n_off), M, granule, vp(pc), vp(seg_pos), vp(p_off), vp(seg_off), vp(points)) == 0
        pair_of_seg = np.repeat(np.arange(M), Ns)
        for l in range(nL):
            w_pc, w_pos, w_off = flat_layout(counts[l], n_off, granule)
            assert np.array_equal(pc[l], w_pc) and np.array_equal(seg_pos[l], w_pos) and np.array_equal(p_off[l], w_off)
            assert np.array_equal(seg_off[l, :S], w_pos + w_off[pair_of_seg]) and np.array_equal(seg_off[l, S:], w_pos)
            assert np.array_equal(points[l], np.add.reduceat(counts[l].astype(np.int64), n_off[:-1]))
    assert lib.sp_host_layout(vp(counts), nL, S, vp(n_off), M, 100, vp(pc), vp(seg_pos), vp(p_off), vp(seg_off), vp(points)) == -1          # SP_EINVAL: not a granule


def test_tensor_list_handles_fast_path_and_conversion():
    """optim.batch_prepare.handles: a list that passes the C-level checks comes back as it is (pointers of the tensors themselves); one
    non-contiguous or non-float32 member sends the list through the per-tensor conversion, and the pointers are those of the copies."""
    import numpy as np
    import torch
    from super_primitive_amd.optim.batch_prepare import handles
    dev = torch.device("cpu")
    good = [torch.arange(9, dtype=torch.float32).reshape(3, 3) + i for i in range(5)]
    ptr, own = handles(good, dev)
    assert own is good and ptr.dtype == np.uint64 and ptr.tolist() == [t.data_ptr() for t in good]
    mixed = list(good)
    mixed[2] = torch.arange(18, dtype=torch.float64).reshape(3, 6)[:, ::2]          # wrong dtype, not contiguous
    ptr2, own2 = handles(mixed, dev)
    assert own2 is not mixed and own2[2].dtype == torch.float32 and own2[2].is_contiguous()
    assert torch.equal(own2[2], mixed[2].float()) and ptr2[2] == own2[2].data_ptr() and ptr2[0] == good[0].data_ptr()


def test_keyframe_drops_its_set_up_record_on_assignment_and_never_copies_it():
    """image.keyframe.KeyFrame: the record the batched set-up keeps on a keyframe (device addresses of its tensors) goes away when one of the
    five tensors is assigned, and is neither deep-copied nor pickled."""
    import copy
    import pickle
    import torch
    from super_primitive_amd.image.keyframe import KeyFrame
    kf = KeyFrame(torch.zeros(3, 4, 4), torch.eye(3), torch.zeros(2, 4, 4), torch.zeros(2, 2), torch.zeros(2, 4, 4, dtype=torch.bool))
    for name in ("image", "K", "logdepth_perseg", "keypoints", "keypoint_regions"):
        kf.__dict__["_sp_prep"] = ("record",)
        setattr(kf, name, getattr(kf, name).clone())
        assert "_sp_prep" not in kf.__dict__, name
    kf.__dict__["_sp_prep"] = ("record",)
    kf.id = 7                                              # any other attribute leaves it alone
    assert "_sp_prep" in kf.__dict__
    assert "_sp_prep" not in copy.deepcopy(kf).__dict__ and "_sp_prep" not in pickle.loads(pickle.dumps(kf)).__dict__
    assert "_sp_prep" in kf.__dict__


def test_schedule_lays_out_three_attempts_and_picks_the_damping_from_the_segments():
    """Host logic of the scheduled run (no GPU): ``PairBatch.schedule`` on a stand-in batch -- the SpSchedule of REFERENCE_START_SCHEDULE
    holds the third attempt's Adam phases in front (retry2_entry = 0, SP_PHASE_ADAM, joining the list at its finest joint phase), then the
    second attempt's pose-only phase (retry_entry, joining at the first joint phase), then the first attempt's list (entry); every phase
    carries SP_PHASE_PREDICTED_EXIT; the damping of the damped coarse phase follows the batch's points per segment."""
    import types
    import torch
    from super_primitive_amd import _lib
    from super_primitive_amd.optim import pair_batch as pb
    t = lambda n=4: torch.zeros(n, dtype=torch.float32)
    lay = lambda pts: types.SimpleNamespace(desc=torch.zeros(8, dtype=torch.uint8), chunks=torch.zeros(4, dtype=torch.int32), spans=torch.zeros(4, dtype=torch.int32),
                                            n_spans=1, partials=t(), seg_partials=t(), points=pts)
    fake = types.SimpleNamespace(level_ids=[0, 1, 2], point_stride={0: 2, 1: 2, 2: 4}, desc={0: torch.zeros(8, dtype=torch.uint8)}, chunks=torch.zeros(4, dtype=torch.int32),
                                 spans=torch.zeros(4, dtype=torch.int32), n_spans=1, partials=t(), seg_partials=t(), wave_flag=_lib.SP_COST_WAVE_SPANS,
                                 table_flag=_lib.SP_COST_DEPTH_TABLE, adam_state=t(64), Ns=[64, 64], Ps=[373056, 373056],
                                 coarse={(0, 2): lay([93264, 93264]), (1, 2): lay([93264, 93264]), (2, 4): lay([23316, 23316])})
    fake.auto_coarse_damping = lambda level, stride: pb.PairBatch.auto_coarse_damping(fake, level, stride)
    kw = {k: v for k, v in pb.REFERENCE_START_SCHEDULE.items() if k != "check_every"}
    s = pb.PairBatch.schedule(fake, **kw)
    n2, n1 = len(pb.REFERENCE_START_ADAM), len(pb.REFERENCE_START_RETRY)
    assert (s.n_phases, s.retry2_entry, s.retry_entry, s.entry) == (n2 + n1 + 6, 0, n2, n2 + n1)
    flags = [s.phase[p].flags for p in range(s.n_phases)]
    assert all(f & _lib.SP_PHASE_ADAM for f in flags[:n2]) and not any(f & _lib.SP_PHASE_ADAM for f in flags[n2:])
    assert all(f & _lib.SP_PHASE_PREDICTED_EXIT for f in flags) and all(f & _lib.SP_PHASE_WAVE_SPANS and f & _lib.SP_PHASE_DEPTH_TABLE for f in flags)
    assert [s.phase[p].max_iters for p in range(n2)] == [500, 500, 500] and abs(s.adam_lr_pose - 1e-2) < 1e-9 and abs(s.adam_lr_kld - 1e-3) < 1e-9
    assert s.adam_state == fake.adam_state.data_ptr()
    first = n2 + n1                                    # pose-only, damped, joint L2, joint L1, joint L0, polish
    assert s.phase[first].flags & _lib.SP_PHASE_POSE_ONLY and s.phase[n2].flags & _lib.SP_PHASE_POSE_ONLY
    assert s.phase[n2 - 1].next == first + 4           # the third attempt joins at the finest joint phase (level 0, stride 2) ...
    assert s.phase[n2 + n1 - 1].next == first + 1      # ... the second at the first phase that is not pose-only (the damped one)
    damp = lambda p: (s.phase[p].flags >> _lib.SP_PHASE_DEPTH_DAMP_SHIFT) & 0xff
    assert damp(first + 1) == 8 * 16 and all(damp(p) == 0 for p in range(s.n_phases) if p != first + 1)      # 364 coarse points per segment: damping 16
    assert s.phase[s.n_phases - 1].spans == fake.spans.data_ptr() and s.phase[first + 4].spans == fake.coarse[(0, 2)].spans.data_ptr()
    # many small segments: a dozen lattice points each -> damping 12
    fake.Ns = [1200, 1200]
    s2 = pb.PairBatch.schedule(fake, **kw)
    assert (s2.phase[first + 1].flags >> _lib.SP_PHASE_DEPTH_DAMP_SHIFT) & 0xff == 8 * 12
    # without later attempts the list stands alone
    s3 = pb.PairBatch.schedule(fake, **dict(kw, retry_phases=None, retry2_phases=None))
    assert (s3.n_phases, s3.entry, s3.retry_entry, s3.retry2_entry, s3.adam_state) == (6, 0, -1, -1, None)
    # only the third attempt (no second): it still joins, and the verdict's thresholds are the documented ones
    s4 = pb.PairBatch.schedule(fake, **dict(kw, retry_phases=None))
    assert (s4.entry, s4.retry_entry, s4.retry2_entry) == (n2, -1, 0) and s4.phase[n2 - 1].next == n2 + 4
    assert pb.VERDICT_DEFAULTS["seg_mean_ratio"] == 1.3 and pb.VERDICT_DEFAULTS["seg_max_ratio"] == 8.0 and pb.VERDICT_DEFAULTS["seg_product"] == 0.9
    assert pb.VERDICT_DEFAULTS["retry_on"] & _lib.SP_STATUS_SEGMENTS
