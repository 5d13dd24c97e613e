"""ctypes binding of libsp_hip.so (the C ABI declared in include/sp_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C super_primitive_amd/csrc``.  There is
no fallback: if the shared object is missing or a symbol cannot be resolved, importing a compute entry point
raises -- the product path never routes through a CPU or pure-PyTorch implementation.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SP_HIP_LIB") or os.path.join(_HERE, "csrc", "libsp_hip.so")   # SP_HIP_LIB: developer A/B builds

P = c_void_p
I = c_int
F = c_float

# name -> argtypes (restype is always int); mirrors include/sp_hip.h one to one
SIGNATURES = {
    "sp_abi_version": [],
    "sp_mask_count": [P, I, I, I, P, P, P, P],
    "sp_table_fill": [P, P, P, I, I, I, P, P, P, P, P, P],
    "sp_table_sample_source": [P, P, P, P, P, I, I, I, I, P, I, I, P, P, I, P],
    "sp_pack_rgb": [P, I, I, I, P, P],
    "sp_blur_decimate": [P, I, I, I, P, P],
    "sp_photo_cost_grad": [P, P, P, P, P, P, I, I, I, I, I, P, P, P, I, I, P, P, I, P, P, F, P, P, P, P, P, P],
    "sp_points_workspace_floats": [I, I],
    "sp_points_cost_grad": [P, P, I, I, I, I, P, I, I, P, P, I, P, P, F, P, P, P, P, P],
    "sp_photo_stats": [P, P, P, P, I, I, I, I, P, P, P, I, I, P, P, I, P, P, F, P, P, P, P, P, P, P, P, I, P],
    "sp_pairs_cost": [P, P, P, I, I, F, P, P, P],
    "sp_pairs_adam_step": [P, I, I, P, P, F, F, F, P, P, P],
    "sp_pairs_gn_step": [P, I, I, P, P, F, F, F, P, P, P, P],
    "sp_pairs_cost_active": [P, P, P, I, I, F, P, P, P, P],
    "sp_pairs_gn_step_conv": [P, I, I, P, P, F, F, F, P, P, P, F, P, P],
    "sp_prepare_count": [P, I, I, I, P],
    "sp_prepare_count_boxed": [P, I, I, I, P],
    "sp_prepare_fill": [P, I, I, I, P],
    "sp_prepare_sample": [P, I, I, P],
    "sp_prepare_sample_pairs": [P, I, P, I, I, P],
    "sp_prepare_blur": [P, I, I, I, P],
    "sp_prepare_pack": [P, I, I, P],
    "sp_prepare_blur_pack": [P, I, I, P],
    "sp_prepare_gather": [P, P, I, P, P],
    "sp_host_work_list_chunks": [P, I, I, I],
    "sp_host_layout": [P, I, I, P, I, I, P, P, P, P, P],
    "sp_host_work_list": [P, P, P, I, I, I, I, I, P, P, P, P, P, P],
    "sp_pairs_schedule_cost": [P, P, P],
    "sp_pairs_schedule_gn_step": [P, I, I, F, F, F, P, P, P, P, P, P, P],
    "sp_pairs_schedule_run": [P, I, I, F, F, F, P, P, P, P, P, I, I, P, P, P, P],
    "sp_pairs_schedule_run_queue": [P, P, I, I, F, F, F, P, P, P, P, P, I, I, P, P, P, P],
    "sp_pairs_adam_iterate": [P, P, P, I, I, I, P, P, P, F, F, F, P, P, P],
    "sp_pairs_gn_iterate": [P, P, P, I, I, I, F, P, P, P, F, F, F, P, P, P, P],
    "sp_window_scratch_doubles": [I, I],
    "sp_window_compose": [P, P, I, P, I, P],
    "sp_window_step": [P, P, I, P, I, P, I, I, P, P, P, I, I, F, P, P, I, P],
    "sp_window_gn_scratch_doubles": [I, I, I, I, I],
    "sp_window_gn_profile_offset": [I, I, I, I, I],
    "sp_window_gn_run": [P, P, P, I, F, P, I, P, I, P, I, I, I, I, P, P, P, P, P, I, F, F, F, F, P, P, I, I, I, P, P],
    "sp_window_gn_multi_bytes": [],
    "sp_window_gn_run_multi": [P, I, F, I, F, F, F, F, I, I, P, P, P, P],
    "sp_window_gn_step": [P, P, I, P, I, P, I, I, I, I, P, P, P, P, P, I, F, F, F, F, P, P, I, P],
    "sp_depth_expand": [P, P, P, P, I, I, I, I, P, P],
    "sp_depth_splat_mean": [P, P, P, P, P, I, I, I, I, P, P, P, P, P],
    "sp_depth_splat": [P, P, P, P, P, I, I, I, I, P, P, P, P, P],
    "sp_segment_reinit": [P, P, P, P, I, I, I, I, P, I, P, P, P, P],
    "sp_depth_average": [P, P, P, P, P, P, I, I, I, I, P, P, P, P],
    "sp_depth_accumulate": [P, P, P, P, P, P, I, I, I, I, P, P],
    "sp_depth_average_finish": [P, I, I, P, P, P],
    "sp_kf_criterion": [P, I, F, P, P, P, P],
    "sp_chain_step": [P, P],
    "sp_kf_criterion_ws_words": [],
    "sp_kf_criterion_ws": [P, I, F, P, P, P, P, P],
    "sp_se3_retract": [P, P, I, P, P, P, P],
    "sp_renormalise_se3": [P, I, P],
    "sp_depth_discontinuity": [P, P, I, I, I, I, F, P, P, P, P],
    "sp_label_components": [P, I, I, I, P, P, P, P],
    "sp_collect_parts": [P, P, P, P, I, I, I, I, P, P, P, P],
    "sp_build_part_masks": [P, P, P, I, I, P, I, P, P],
    "sp_kth_mask_pixel": [P, P, I, I, I, P, P, P],
}

SP_ABI_VERSION = 15
SP_GRAD_PARTIAL_FLOATS = 16
SP_GN_PARTIAL_FLOATS = 32
SP_GNA_PARTIAL_FLOATS = 48
SP_GNA_SEG_FLOATS = 12
SP_GRAD_SEG_FLOATS = 1
SP_GN_SEG_FLOATS = 12
SP_LM_STATE_FLOATS = 8


class SpPair(ctypes.Structure):
    """Mirror of ``struct SpPair`` (include/sp_hip.h); 136 bytes."""
    _fields_ = [
        ("pix", c_void_p), ("src4", c_void_p), ("kp_L", c_void_p), ("trg3", c_void_p),
        ("kld", c_void_p), ("pose", c_void_p), ("aff", c_void_p), ("seg_tile_off", c_void_p),
        ("K_src", c_float * 4), ("K_trg", c_float * 4),
        ("N", c_int), ("P", c_int), ("H", c_int), ("W", c_int), ("Hl", c_int), ("Wl", c_int),
        ("tile0", c_int), ("n_tiles", c_int), ("zmin", c_float), ("rec0", c_int),
    ]


SP_MAX_PHASES = 12
SP_PHASE_POSE_ONLY = 1
SP_PHASE_WAVE_SPANS = 2
SP_COST_WAVE_SPANS = 0x100
SP_COST_DEPTH_TABLE = 0x200
SP_PHASE_DEPTH_TABLE = 4
SP_PHASE_ADAM = 8
SP_PHASE_PREDICTED_EXIT = 16
SP_PHASE_DEPTH_DAMP_SHIFT = 8
SP_PREP_DEPTH_TABLE = 0x10000
SP_PREP_DENSE_L = 0x20000


SP_PREP_MAX_STRIDES = 4
SP_PREP_MAX_LEVELS = 4


class SpPrepTable(ctypes.Structure):
    """Mirror of ``struct SpPrepTable`` (include/sp_hip.h): one keyframe of the batched preparation, all its lattices."""
    _fields_ = [("masks", c_void_p), ("logdepth", c_void_p), ("keypoints", c_void_p), ("kp_L", c_void_p),
                ("row_counts", c_void_p * SP_PREP_MAX_STRIDES), ("counts", c_void_p * SP_PREP_MAX_STRIDES),
                ("seg_off", c_void_p * SP_PREP_MAX_STRIDES), ("pix", c_void_p * SP_PREP_MAX_STRIDES),
                ("baseL", c_void_p * SP_PREP_MAX_STRIDES), ("stride", c_int * SP_PREP_MAX_STRIDES),
                ("N", c_int), ("H", c_int), ("W", c_int), ("n_strides", c_int), ("bits", c_void_p), ("boxes", c_void_p)]


class SpPrepSample(ctypes.Structure):
    _fields_ = [("pix", c_void_p), ("baseL", c_void_p), ("seg_off", c_void_p), ("counts", c_void_p), ("kp_L", c_void_p), ("kld", c_void_p),
                ("K", c_void_p), ("image", c_void_p * SP_PREP_MAX_LEVELS), ("src4", c_void_p * SP_PREP_MAX_LEVELS),
                ("Hl", c_int * SP_PREP_MAX_LEVELS), ("Wl", c_int * SP_PREP_MAX_LEVELS),
                ("N", c_int), ("P", c_int), ("H", c_int), ("W", c_int), ("n_levels", c_int), ("granule", c_int)]


class SpPrepImage(ctypes.Structure):
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("H", c_int), ("W", c_int)]


class SpPrepImagePack(ctypes.Structure):
    """Mirror of ``struct SpPrepImagePack`` (include/sp_hip.h): a pyramid step with the packed forms of its input / output level; 40 bytes."""
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("packed_in", c_void_p), ("packed_out", c_void_p), ("H", c_int), ("W", c_int)]


class SpPhase(ctypes.Structure):
    """Mirror of ``struct SpPhase`` (include/sp_hip.h)."""
    _fields_ = [("pairs", c_void_p), ("chunks", c_void_p), ("spans", c_void_p), ("span_partials", c_void_p), ("seg_partials", c_void_p),
                ("n_spans", c_int), ("max_iters", c_int), ("irls_eps", c_float), ("conv_tol", c_float), ("flags", c_int), ("next", c_int)]


class SpSchedule(ctypes.Structure):
    """Mirror of ``struct SpSchedule``; lives in host memory, passed by address (``ctypes.addressof``)."""
    _fields_ = [("phase", SpPhase * SP_MAX_PHASES), ("n_phases", c_int), ("entry", c_int), ("retry_entry", c_int), ("retry2_entry", c_int),
                ("adam_lr_pose", c_float), ("adam_lr_kld", c_float), ("adam_state", c_void_p)]


SP_STATUS_NONFINITE = 1
SP_STATUS_LAST_CAP = 2
SP_STATUS_DEPTH_RANGE = 4
SP_STATUS_COST = 8
SP_STATUS_VALID = 16
SP_STATUS_SEGMENTS = 32
SP_STATUS_RETRIED = 0x100
SP_STATUS_UNFINISHED = 0x200
SP_STATUS_ADAM = 0x400
SP_STATUS_FAILED = SP_STATUS_NONFINITE | SP_STATUS_LAST_CAP | SP_STATUS_DEPTH_RANGE | SP_STATUS_COST | SP_STATUS_VALID | SP_STATUS_SEGMENTS | SP_STATUS_UNFINISHED
SP_VERDICT_SEGMENTS = 2048
SP_VERDICT_SEGMENT_POINTS = 64
SP_VERDICT_MIN_SEGMENTS = 8
SP_DIAG_FLOATS = 8


class SpVerdict(ctypes.Structure):
    """Mirror of ``struct SpVerdict`` (include/sp_hip.h): the per-pair verdict of a scheduled run and its later attempts."""
    _fields_ = [("status", c_void_p), ("diag", c_void_p), ("attempts", c_void_p), ("pose0", c_void_p), ("kld0", c_void_p),
                ("pose_base", c_void_p), ("kld_base", c_void_p), ("kld_bound", c_float), ("cost_bound", c_float), ("cost_ratio", c_float),
                ("valid_min", c_float), ("retry_mask", c_int), ("lam0", c_float), ("seg_max_ratio", c_float), ("seg_mean_ratio", c_float), ("evals", c_void_p),
                ("seg_product", c_float), ("pad_", c_float)]


class SpQueue(ctypes.Structure):
    """Mirror of ``struct SpQueue`` (include/sp_hip.h): slot-level continuous batching of a scheduled run; host memory."""
    _fields_ = [("qpairs", c_void_p * SP_MAX_PHASES), ("slot_pairs", c_void_p * SP_MAX_PHASES), ("max_spans", c_int * SP_MAX_PHASES),
                ("n_queue", c_int), ("pad_", c_int),
                ("head", c_void_p), ("slot_pair", c_void_p), ("q_costs", c_void_p), ("q_lm", c_void_p), ("lam0", c_float), ("pad2_", c_int),
                ("active", c_void_p)]


class SpWindowGn(ctypes.Structure):
    """Mirror of ``struct SpWindowGn`` (include/sp_hip.h): one window of a multi-window Gauss-Newton run; 136 bytes."""
    _fields_ = [("pairs", c_void_p), ("chunks", c_void_p), ("spans", c_void_p), ("edges", c_void_p), ("nodes", c_void_p), ("blocks", c_void_p),
                ("span_partials", c_void_p), ("seg_partials", c_void_p), ("scratch", c_void_p), ("nodes_backup", c_void_p), ("kld_backup", c_void_p),
                ("state", c_void_p), ("losses", c_void_p), ("n_spans", c_int), ("n_edges", c_int), ("n_nodes", c_int), ("n_blocks", c_int),
                ("sum_N", c_int), ("max_N", c_int), ("n_unknowns", c_int), ("max_losses", c_int)]


SP_CHAIN_LEVELS, SP_CHAIN_PHASES = 4, 6
SP_CHAIN_TRACK, SP_CHAIN_SUPP, SP_CHAIN_CRITERION = 1, 2, 4


class SpChainPhase(ctypes.Structure):
    _fields_ = [("level", c_int), ("max_iters", c_int), ("irls_eps", c_float), ("conv_tol", c_float)]


class SpChainWindow(ctypes.Structure):
    """Mirror of ``struct SpChainWindow`` (include/sp_hip.h): a built window with its Gauss-Newton schedule; 680 bytes."""
    _fields_ = [("gn", SpWindowGn * SP_CHAIN_LEVELS), ("phase", SpChainPhase * SP_CHAIN_PHASES), ("n_phases", c_int), ("check_every", c_int),
                ("flags", c_int), ("lam0", c_float), ("lm_up", c_float), ("lm_down", c_float), ("lm_min", c_float), ("check_first", c_int),
                ("state_host", c_void_p)]


class SpChainTarget(ctypes.Structure):
    """Mirror of ``struct SpChainTarget``; 56 bytes."""
    _fields_ = [("packed", c_void_p * SP_CHAIN_LEVELS), ("pose", c_void_p), ("aff", c_void_p), ("node", c_int), ("pad_", c_int)]


class SpChainStep(ctypes.Structure):
    """Mirror of ``struct SpChainStep`` (include/sp_hip.h): the arguments of one frame of the odometry chain."""
    _fields_ = [("stages", c_int), ("H", c_int), ("W", c_int), ("n_levels", c_int), ("image", c_void_p), ("level", c_void_p * SP_CHAIN_LEVELS),
                ("track", SpChainWindow), ("track_target", SpChainTarget), ("out_pose", c_void_p), ("out_aff", c_void_p),
                ("supp", SpChainWindow), ("supp_target", SpChainTarget * 2), ("supp_images", c_int), ("kld_n", c_int), ("kld_src", c_void_p),
                ("kld_dst", c_void_p),
                ("pix", c_void_p), ("baseL", c_void_p), ("seg_off", c_void_p), ("kp_L", c_void_p), ("kld", c_void_p), ("K", c_void_p),
                ("kf_pose", c_void_p), ("N", c_int), ("P", c_int), ("keys", c_void_p), ("depth_out", c_void_p), ("rel_pose", c_void_p),
                ("crit", c_void_p), ("crit_ws", c_void_p), ("crit_host", c_void_p), ("valid_thresh", c_float), ("track_iters", c_int), ("supp_iters", c_int),
                ("pad_", c_int)]


class SpWindowNode(ctypes.Structure):
    """Mirror of ``struct SpWindowNode`` (include/sp_hip.h); 176 bytes."""
    _fields_ = [("T", c_float * 16), ("a", c_float * 6), ("m", c_float * 6), ("v", c_float * 6), ("aff", c_float * 2),
                ("aff_m", c_float * 2), ("aff_v", c_float * 2), ("lr_pose", c_float), ("lr_aff", c_float),
                ("kind", c_int), ("flags", c_int)]


class SpWindowEdge(ctypes.Structure):
    _fields_ = [("src_node", c_int), ("trg_node", c_int), ("block", c_int), ("weight", c_float)]


class SpWindowBlock(ctypes.Structure):
    _fields_ = [("kld", c_void_p), ("m", c_void_p), ("v", c_void_p), ("N", c_int), ("lr", c_float)]


_lib = None


def load():
    """Load (once) and type the library.  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the hot path.")
    try:
        import torch  # noqa: F401  -- make sure torch's HIP runtime (libamdhip64.so.7) is the one already mapped
    except Exception:  # pragma: no cover - torch is a hard dependency of the package anyway
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = c_int
    if lib.sp_abi_version() != SP_ABI_VERSION:
        raise RuntimeError("libsp_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "SP_EINVAL (bad argument)", -2: "SP_ELIMIT (size not supported)"}.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"{what} failed: {kind}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = None


def stream_ptr():
    """The current device's current HIP stream (what ``torch.cuda.current_stream().cuda_stream`` returns, without building the
    Stream object: 0.4 us instead of 10 -- a tracked frame of the config-3 chain asks twenty times).  The fast path uses two private
    torch entry points, resolved ONCE; a torch that no longer has them falls back to the public API."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        raw, cur = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if raw is not None and cur is not None:
            _raw_stream = lambda: raw(cur())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return ctypes.c_void_p(_raw_stream())


def require_device(*tensors):
    """The hot path runs on the GPU only; refuse host tensors instead of silently computing elsewhere.

    The raw launches go to the CURRENT device's current stream with bare pointers, so -- unlike ATen ops, which
    switch device under the hood -- operands on different devices, or on a device that is not the current one,
    would fault or silently rely on peer access: both are refused here."""
    import torch
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("super_primitive_amd: the photometric hot path is HIP-only; got a CPU tensor. "
                               "Move the keyframe to a cuda device (no CPU fallback exists).")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"super_primitive_amd: operands on different devices ({dev} and {t.device})")
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"super_primitive_amd: operands live on {dev} but the current device is "
                           f"cuda:{torch.cuda.current_device()}; wrap the call in `with torch.cuda.device({dev.index}):` "
                           "(one process per GPU is the supported layout)")
