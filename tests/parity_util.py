"""Error measures shared by the parity tests and tools/parity_report.py (north_star bar: pose within 1e-4 rad / 1e-4 t,
depth within 1e-3 relative of the reference)."""
import hashlib

import numpy as np


def rot_angle(Ra, Rb):
    """Angle of Ra^T Rb from the skew part (accurate near zero, where arccos of an fp32 trace is not)."""
    R = np.asarray(Ra, np.float64)[:3, :3].T @ np.asarray(Rb, np.float64)[:3, :3]
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    c = 0.5 * (np.trace(R) - 1.0)
    return float(np.arctan2(s, c))


def pose_depth_errors(pose, kld, pose_ref, kld_ref, gauge=True):
    """(rotation [rad], translation [max abs], depth [max relative]) of (pose, kld) against the reference's.  gauge=True
    first removes the one unobservable degree of freedom of a two-view problem, the global scale (t -> s t,
    kld -> kld + log s leaves the photometric cost unchanged): s = exp(mean(kld_ref - kld))."""
    pose, pose_ref = np.asarray(pose, np.float64), np.asarray(pose_ref, np.float64)
    kld, kld_ref = np.asarray(kld, np.float64), np.asarray(kld_ref, np.float64)
    ls = float(np.mean(kld_ref - kld)) if gauge else 0.0
    t = pose[:3, 3] * np.exp(ls)
    return (rot_angle(pose, pose_ref), float(np.abs(t - pose_ref[:3, 3]).max()),
            float(np.abs(np.expm1(kld + ls - kld_ref)).max()))


def rel_max(got, want):
    """max |got - want| / max |want| (gradient entries span orders of magnitude)."""
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-300))


def input_digest(pair):
    """sha256 over the arrays of a synthetic pair, as stored by oracle/gen_goldens_fullsize.py (``in_sha256``)."""
    h = hashlib.sha256()
    for a in (pair.src_image, pair.trg_image, pair.K, pair.logdepth_perseg, pair.keypoints,
              np.packbits(pair.keypoint_regions, axis=-1), pair.kld_init, pair.pose_init):
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def fullsize_pair(g):
    """Regenerate the synthetic pair a full-size golden was recorded on and check it is bit-identical to the recorded one."""
    from super_primitive_amd import synth
    return pair_from_args(str(g["make_pair_args"]), int(g["seed"]), g["in_sha256"])


def pair_from_args(make_pair_args, seed, digest=None):
    """``synth.make_pair`` from a golden's recorded argument string ("H=..,W=..,N=..,overlap=..,init_sigma=..[,texture=..,
    init_mode=..]"), checked against the recorded input digest."""
    from super_primitive_amd import synth
    kw = dict(item.split("=") for item in make_pair_args.split(","))
    extra = {k: kw[k] for k in ("texture", "init_mode", "shape") if k in kw}
    pair = synth.make_pair(int(kw["H"]), int(kw["W"]), int(kw["N"]), seed=int(seed), overlap=int(kw.get("overlap", 0)),
                           init_sigma=float(kw["init_sigma"]), **extra)
    if digest is not None:
        assert np.array_equal(input_digest(pair), digest), "regenerated inputs differ from the ones the golden was recorded on"
    return pair
