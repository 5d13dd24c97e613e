"""The oracle (oracle/photometric_oracle.py) replayed against vectors produced by the REAL reference
(oracle/gen_goldens.py).  CPU only.  This is what pins the oracle; the GPU tests then pin the HIP path
to the oracle and to the same vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden, unpack_masks
from oracle import photometric_oracle as orc

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def frames(g, src_img=None, trg_img=None, K_img=None):
    masks = unpack_masks(g)
    src = orc.OracleFrame(T(g["in_src_image"] if src_img is None else src_img), T(g["in_K"]), T(g["in_logdepth"]),
                          T(g["in_keypoints"]), T(masks), K_img=None if K_img is None else T(K_img))
    trg = orc.OracleFrame(T(g["in_trg_image"] if trg_img is None else trg_img), T(g["in_K"]),
                          K_img=None if K_img is None else T(K_img))
    return src, trg


COST_CASES = ["g1_grid_48x64", "g1_blobs_affine_60x80", "g1_pyramid_72x96", "g1_behind_camera_48x64", "g1_odd_45x67"]


@pytest.mark.parametrize("name", COST_CASES)
def test_cost_forward_stats_and_grads(name):
    g = load_golden(name)
    for li in range(int(g["n_levels"])):
        p = f"L{li}_"
        src, trg = frames(g, g[p + "lvl_src_image"], g[p + "lvl_trg_image"], g[p + "lvl_K_img"])
        kld = T(g["in_kld"]).requires_grad_(True)
        pose = T(g["in_pose"]).requires_grad_(True)
        aff = None
        if "in_aff_src" in g:
            aff = (T(g["in_aff_src"]).requires_grad_(True), T(g["in_aff_trg"]).requires_grad_(True))
        out = orc.photometric_cost(src, trg, kld, pose, collect_stats=2, affine=aff)
        out["residual"].abs().mean().backward()
        np.testing.assert_allclose(out["residual"].detach().numpy(), g[p + "residual"], rtol=1e-6, atol=1e-9)
        for key in ("segm_ids", "src_valid_mask", "trg_valid_mask", "full_mask", "src_in_trg_keypoints_valid_mask"):
            assert np.array_equal(out[key].numpy(), g[p + key]), key
        for key in ("src_pixels", "src_in_trg_pixels", "src_pts", "src_in_trg_pts", "residual_raw",
                    "src_in_trg_keypoints", "src_in_trg_keypoints_z"):
            np.testing.assert_allclose(out[key].detach().numpy(), g[p + key], rtol=1e-5, atol=1e-6, err_msg=key)
        np.testing.assert_allclose(kld.grad.numpy(), g[p + "g_kld"], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(pose.grad.numpy(), g[p + "g_pose"], rtol=1e-4, atol=1e-7)
        if aff is not None:
            np.testing.assert_allclose(aff[0].grad.numpy(), g[p + "g_aff_src"], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(aff[1].grad.numpy(), g[p + "g_aff_trg"], rtol=1e-4, atol=1e-8)


def test_behind_camera_case_really_has_invalid_points():
    g = load_golden("g1_behind_camera_48x64")
    z = g["L0_src_in_trg_pts"][:, 2]
    assert (z < 0).any() and (z > 1e-7).any()
    assert 0 < g["L0_trg_valid_mask"].sum() < g["L0_trg_valid_mask"].size


def test_precomputed():
    g = load_golden("g3_precomputed_60x80")
    src, trg = frames(g)
    with torch.no_grad():
        pre = orc.unproject_keyframe(src, T(g["in_kld"]))
    assert np.array_equal(pre["segm_ids"].numpy(), g["pre_segm_ids"])
    assert np.array_equal(pre["src_valid_mask"].numpy(), g["pre_src_valid_mask"])
    np.testing.assert_allclose(pre["src_pts"].numpy(), g["pre_src_pts"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(pre["src_pixels"].numpy(), g["pre_src_pixels"], rtol=1e-5, atol=1e-6)
    assert tuple(pre["spatial_size"]) == tuple(g["pre_spatial_size"])
    pose = T(g["in_pose"]).requires_grad_(True)
    a0 = T(g["in_aff_src"]).requires_grad_(True)
    a1 = T(g["in_aff_trg"]).requires_grad_(True)
    out = orc.photometric_cost_precomputed(pre, trg, pose, affine=(a0, a1))
    out["residual"].mean().backward()
    np.testing.assert_allclose(out["residual"].detach().numpy(), g["residual"], rtol=1e-6)
    np.testing.assert_allclose(pose.grad.numpy(), g["g_pose"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(a0.grad.numpy(), g["g_aff_src"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(a1.grad.numpy(), g["g_aff_trg"], rtol=1e-4, atol=1e-8)


def test_batch():
    g = load_golden("g4_batch3_48x64")
    src, _ = frames(g)
    B = g["in_poses"].shape[0]
    kld = T(g["in_kld"]).requires_grad_(True)
    P = T(g["in_poses"]).requires_grad_(True)
    a0 = T(g["in_aff_src"]).requires_grad_(True)
    a1 = T(g["in_aff_trg"]).requires_grad_(True)
    out = orc.photometric_cost_batch(src, T(g["in_trg_images"]), T(g["in_trg_Ks"]), kld, P, collect_stats=1, affine=(a0, a1))
    (out["residual"] * torch.arange(1, B + 1, dtype=torch.float32)).sum().backward()
    np.testing.assert_allclose(out["residual"].detach().numpy(), g["residual"], rtol=1e-6)
    assert np.array_equal(out["full_mask"].numpy(), g["full_mask"])
    np.testing.assert_allclose(out["src_in_trg_pixels"].detach().numpy(), g["src_in_trg_pixels"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(kld.grad.numpy(), g["g_kld"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(P.grad.numpy(), g["g_pose"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(a0.grad.numpy(), g["g_aff_src"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(a1.grad.numpy(), g["g_aff_trg"], rtol=1e-4, atol=1e-8)


def test_pyramid_levels():
    g = load_golden("g5_pyramid")
    for tag in ("even", "odd"):
        f = orc.OracleFrame(T(g[f"{tag}_image"]), T(g[f"{tag}_Kin"]))
        for (s, e) in ((0, 3), (1, 4), (0, 1)):
            pyr = orc.frame_pyramid(f, s, e)
            assert len(pyr) == int(g[f"{tag}_{s}_{e}_n"])
            for i, k in enumerate(pyr):
                ref = g[f"{tag}_{s}_{e}_img{i}"]
                assert tuple(k.image.shape) == ref.shape
                np.testing.assert_allclose(k.image.numpy(), ref, rtol=0, atol=2e-7)
                np.testing.assert_allclose(k.K_img.numpy(), g[f"{tag}_{s}_{e}_Kimg{i}"], rtol=1e-7)
                np.testing.assert_allclose(k.K.numpy(), g[f"{tag}_{s}_{e}_K{i}"], rtol=0)


def test_depth_render():
    g = load_golden("g6_depth_render")
    masks = unpack_masks(g)
    f = orc.OracleFrame(T(g["in_src_image"]), T(g["in_K"]), T(g["in_L_const"]), T(g["in_keypoints"]), T(masks))
    half = orc.render_depth(f, T(g["in_kld_const"]), T(g["in_pose_half"]))
    np.testing.assert_allclose(half.numpy(), g["out_half"], rtol=1e-6)
    ident = orc.render_depth(f, T(g["in_kld_levels"]))
    assert (np.isclose(ident.numpy(), g["out_identity"], rtol=1e-6)).mean() > 0.995
    f2 = orc.OracleFrame(T(g["in_src_image"]), T(g["in_K"]), T(g["in_logdepth"]), T(g["in_keypoints"]), T(masks))
    gen = orc.render_depth(f2, T(g["in_kld_gt"]), T(g["in_pose_gt"]))
    assert (np.isclose(gen.numpy(), g["out_general"], rtol=1e-5)).mean() > 0.99


def test_segment_stats():
    g = load_golden("g7_segment_stats")
    src, _ = frames(g)
    for mode in ("mean", "median"):
        kld, seen = orc.segment_depth_reinit(T(g["in_sparse_depth"]), src, mode)
        assert np.array_equal(seen.numpy(), g[f"{mode}_visible"])
        assert not seen.all()
        np.testing.assert_allclose(kld.numpy(), g[f"{mode}_kld"], rtol=1e-5, atol=1e-6)
    kld, seen = T(g["median_kld"]), T(g["median_visible"])
    dense = orc.keyframe_depths(src, kld)
    np.testing.assert_allclose(float(dense.double().sum()), float(g["depths_dense_sum"]), rtol=1e-9)
    d = dense.clone()
    d[~src.keypoint_regions] = -1
    avg, invalid = orc.average_depths(d[seen])
    assert np.array_equal(invalid.numpy(), g["avg_invalid"])
    np.testing.assert_allclose(avg.numpy(), g["avg_depth"], rtol=1e-6, atol=1e-7)


def test_kf_criteria():
    from oracle import kf_oracle
    g = load_golden("g11_kf_criteria")
    for tag in ("odd", "even", "dense", "one"):
        a, b, d = T(g[f"{tag}_pose_src"]), T(g[f"{tag}_pose_trg"]), T(g[f"{tag}_depth"])
        diff, scale = kf_oracle.translation_difference(a, b, d)
        assert float(scale) == float(g[f"{tag}_scale"])                      # an element of the image: exact
        np.testing.assert_allclose(float(diff), float(g[f"{tag}_diff"]), rtol=1e-6)
        np.testing.assert_allclose(kf_oracle.rotation_difference(a, b), float(g[f"{tag}_angle_deg"]), rtol=1e-9, atol=1e-12)
        assert int((d > 1e-6).sum()) == int(g[f"{tag}_n_valid"])


def test_lie():
    g = load_golden("g8_lie")
    np.testing.assert_allclose(orc.renormalise_se3(T(g["in_noisy"]).clone()).numpy(), g["renorm"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.invert_se3(T(g["in_T"])).numpy(), g["inverse"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.pose_to_tq(T(g["in_T"])).numpy(), g["tq"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.pose_to_tq(T(g["in_T"][1])).numpy(), g["tq_single"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.quat_wxyz_to_R(T(g["in_quat"])).numpy(), g["quat_R"], rtol=1e-6, atol=1e-7)
    for i in range(4):
        np.testing.assert_allclose(orc.se3_log_reference_quirk(T(g["in_T"][i:i + 1])).numpy()[0], g["logmap"][i],
                                   rtol=1e-5, atol=1e-6)


def test_se3_exp_matches_matrix_exponential():
    """lietorch is not under /root/reference (parity unpinned there): the closed form is checked against
    scipy.linalg.expm of the 4x4 twist instead."""
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    for scale in (0.0, 1e-6, 1e-3, 0.3, 2.0):
        xi = rng.standard_normal(6) * scale
        M = np.zeros((4, 4))
        M[:3, :3] = np.array([[0, -xi[5], xi[4]], [xi[5], 0, -xi[3]], [-xi[4], xi[3], 0]])
        M[:3, 3] = xi[:3]
        got = orc.se3_exp(torch.from_numpy(xi)[None])[0].numpy()
        np.testing.assert_allclose(got, expm(M), rtol=1e-9, atol=1e-12)


# ---- G9: K-step Adam trajectories of the three caller loop shapes, replayed with the oracle cost ------------
def test_traj_sfm_loop():
    g = load_golden("g9a_traj_sfm")
    src, trg = frames(g)
    sp, tp = orc.frame_pyramid(src, 0, 2), orc.frame_pyramid(trg, 0, 2)
    kld = torch.nn.Parameter(T(g["in_kld"]).clone())
    a = torch.nn.Parameter(torch.zeros(1, 6))
    T0 = T(g["in_pose_init"])
    opt = torch.optim.Adam([{"params": kld, "lr": 1e-3}, {"params": [a], "lr": 1e-2}], lr=1e-3)
    losses, count = [], 0
    for s, t in zip(sp, tp):
        for _ in range(int(g["steps"])):
            out = orc.photometric_cost(s, t, kld, orc.se3_exp(a)[0] @ T0)
            loss = out["residual"].abs().mean()
            losses.append(float(loss.detach()))
            if count > 0:
                loss.backward()
                opt.step()
                opt.zero_grad()
            count += 1
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3)
    np.testing.assert_allclose(kld.detach().numpy(), g["final_kld"], atol=1e-4)
    np.testing.assert_allclose((orc.se3_exp(a.detach())[0] @ T0).numpy(), g["final_pose"], atol=1e-4)


def test_traj_tracking_loop():
    g = load_golden("g9b_traj_track")
    src, trg = frames(g)
    with torch.no_grad():
        pre = orc.unproject_keyframe(src, T(g["in_kld"]))
    delta = torch.nn.Parameter(torch.zeros(1, 6))
    aff = torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.Adam([{"params": [delta], "lr": 5e-3}, {"params": [aff], "lr": 5e-3}], lr=5e-3)
    supp_T = orc.invert_se3(T(g["in_pose_init"]))
    losses = []
    for _ in range(int(g["steps"])):
        pose = orc.se3_exp(delta)[0] @ orc.invert_se3(supp_T) @ torch.eye(4)
        out = orc.photometric_cost_precomputed(pre, trg, pose, affine=(torch.zeros(2), aff))
        loss = out["residual"].mean()
        losses.append(float(loss.detach()))
        loss.backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            supp_T = supp_T @ orc.invert_se3(orc.se3_exp(delta.detach())[0])
            delta.data.zero_()
    supp_T = orc.renormalise_se3(supp_T)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3)
    np.testing.assert_allclose(supp_T.numpy(), g["final_supp_T"], atol=1e-4)
    np.testing.assert_allclose(aff.detach().numpy(), g["final_aff"], atol=1e-4)


# ---- G10: keyframe post-processing (N2) -- frontend oracle vs the real reference module (cupy stubbed by scipy) ----
def _g10_case(g, tag):
    H, W, N = (int(v) for v in g[f"{tag}_HWN"])
    unpack = lambda a, n: np.unpackbits(a, axis=-1, count=W).astype(bool).reshape(n, H, W)
    return H, W, N, unpack


@pytest.mark.parametrize("tag", ["grid", "blobs"])
def test_post_process_oracle(tag):
    from oracle import frontend_oracle as fo
    g = load_golden("g10_post_process")
    H, W, N, unpack = _g10_case(g, tag)
    masks = T(unpack(g[f"{tag}_masks"], N))
    L = T(g[f"{tag}_L"])
    disc, split = fo.discontinuity(L.clone(), masks)
    assert np.array_equal(disc.numpy(), unpack(g[f"{tag}_disc"], N))
    assert np.array_equal(split.numpy(), unpack(g[f"{tag}_split"], N))
    assert disc.any() and split.any()
    labels, n_lab = fo.label_slices(split)
    assert n_lab == int(g[f"{tag}_n_labels"]) and np.array_equal(labels, g[f"{tag}_labels"])
    frame = orc.OracleFrame(torch.zeros(3, H, W), torch.eye(3), L, T(g[f"{tag}_keypoints"]), masks)
    torch.manual_seed(123)
    nm, nL, nk = fo.fix_disconnected(frame)
    K = int(g[f"{tag}_new_K"])
    assert nm.shape[0] == K and K > N
    assert np.array_equal(nm.numpy(), unpack(g[f"{tag}_new_masks"], K))
    np.testing.assert_allclose(nL.double().sum((1, 2)).numpy(), g[f"{tag}_new_logdepth_sum"], rtol=1e-12)
    np.testing.assert_allclose(nk.numpy(), g[f"{tag}_new_keypoints"], rtol=0, atol=1e-7)
