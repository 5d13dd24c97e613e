"""-m gpu: keyframe post-processing on the HIP path against golden vectors produced by the
real reference module (frontend/segment/post_processer.py with cupy stubbed by scipy) and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_util import T, npy

pytestmark = pytest.mark.gpu


def case(g, tag):
    H, W, N = (int(v) for v in g[f"{tag}_HWN"])
    unpack = lambda a, n: np.unpackbits(a, axis=-1, count=W).astype(bool).reshape(n, H, W)
    return H, W, N, unpack


@pytest.mark.parametrize("tag", ["grid", "blobs"])
def test_post_process_matches_reference(tag):
    from super_primitive_amd.frontend.segment import post_processer as pp
    from super_primitive_amd.image.keyframe import KeyFrame
    g = load_golden("g10_post_process")
    H, W, N, unpack = case(g, tag)
    masks, L = T(unpack(g[f"{tag}_masks"], N)), T(g[f"{tag}_L"])
    disc = pp.depth_discontinuity(L, masks)
    split = pp.mask_by_depth_discontinuity(L, masks)
    # gradient magnitudes exactly at the 0.1 threshold may round differently: allow a couple of pixels
    assert (npy(disc) != unpack(g[f"{tag}_disc"], N)).sum() <= 2
    assert (npy(split) != unpack(g[f"{tag}_split"], N)).sum() <= 2
    # labelling is exact given the same input: feed the reference's split mask
    ref_split = T(unpack(g[f"{tag}_split"], N))
    lab, n_lab = pp.connected_components_batch(ref_split)
    assert n_lab == int(g[f"{tag}_n_labels"])
    assert np.array_equal(lab, g[f"{tag}_labels"])
    # full pipeline with the reference's RNG seed
    kf = KeyFrame(torch.zeros(3, H, W, device=L.device), torch.eye(3, device=L.device), L, T(g[f"{tag}_keypoints"]), masks)
    torch.manual_seed(123)
    new = pp.kf_fix_disconnected_regions(kf)
    K = int(g[f"{tag}_new_K"])
    assert new.keypoint_regions.shape[0] == K
    want = unpack(g[f"{tag}_new_masks"], K)
    assert (npy(new.keypoint_regions) != want).sum() <= 4
    np.testing.assert_allclose(npy(new.logdepth_perseg.double().sum((1, 2))), g[f"{tag}_new_logdepth_sum"], rtol=1e-6)
    np.testing.assert_allclose(npy(new.keypoints), g[f"{tag}_new_keypoints"], rtol=0, atol=1e-6)
    assert kf.keypoint_regions.shape[0] == N, "the input keyframe must be left untouched"
    # every new keypoint lies inside its own mask
    rc = np.round(0.5 * (np.array([H, W]) - 1) * (npy(new.keypoints) + 1)).astype(int)
    m = npy(new.keypoint_regions)
    assert all(m[k, rc[k, 0], rc[k, 1]] for k in range(K))


def test_labelling_matches_scipy_on_random_masks_and_is_a_partition():
    """Random 55 % density masks (many tiny components, long snakes) at 96 x 128 x 10, against the oracle."""
    from oracle import frontend_oracle as fo
    from super_primitive_amd.frontend.segment import post_processer as pp
    rng = np.random.default_rng(5)
    m = rng.uniform(size=(10, 96, 128)) < 0.55
    m[3] = False                          # an empty slice
    m[4] = True                           # a full slice (no background label)
    m[5, ::2] = True; m[5, 1::2] = False  # horizontal stripes: one component per row
    lab, n = pp.connected_components_batch(T(m))
    want, n_want = fo.label_slices(torch.from_numpy(m))
    assert n == n_want and np.array_equal(lab, want)
    assert np.array_equal(lab > 0, m)


def test_full_size_post_process_consistency():
    """640 x 480 x 64 with depth steps: the new segments partition (mask minus nothing) of every kept segment, the
    table of the new keyframe builds, and the photometric cost runs on it."""
    from super_primitive_amd import synth
    from super_primitive_amd.core import dense_optim
    from super_primitive_amd.frontend.segment import post_processer as pp
    from super_primitive_amd.image.keyframe import KeyFrame
    pair = synth.make_pair(480, 640, 64, seed=9, overlap=4)
    L = synth.stepped_logdepth(pair, seed=2, n_boxes=5)
    kf = KeyFrame(T(pair.src_image), T(pair.K), T(L), T(pair.keypoints), T(pair.keypoint_regions))
    torch.manual_seed(0)
    new = pp.kf_fix_disconnected_regions(kf)
    K = new.keypoint_regions.shape[0]
    assert K > 64
    union_new = new.keypoint_regions.any(0)
    union_old = kf.keypoint_regions.any(0)
    assert bool((union_new <= union_old).all())
    # parts of one original segment are pairwise disjoint
    sizes = new.keypoint_regions.sum((1, 2))
    assert int(sizes.min()) > 1e-3 * 480 * 640 - 1
    kld = torch.log(torch.full((K,), 3.0, device=new.keypoints.device))
    trg = KeyFrame(T(pair.trg_image), T(pair.K))
    out = dense_optim.photomeric_cost(new, trg, kld, T(pair.pose_init), {"mode": "colour", "collect_stats": 0})
    assert np.isfinite(npy(out["residual"])).all()
