"""One foreign call per FRAME of the odometry chain (``sp_chain_step``, include/sp_hip.h): the host side that binds the per-keyframe
Gauss-Newton windows of ``loops.GnTracker`` / ``loops.GnSuppMapper`` to the call's argument record.

The reference's driver loop (``odometery/odometery.py:1018-1075``) runs, for a frame that is not a keyframe, ``track_frame`` (:323-449),
``mapping(mode='supp')`` (:1038-1042) and ``is_kf`` (:986-1016).  ``odometery/sequence.py`` mirrors that loop step by step in Python;
with the windows built once per keyframe the interpreter BETWEEN the launches is most of a frame (0.76 ms of kernels
in 2.7 ms).  ``ChainStep`` keeps poses, affine pairs and depths on the device and hands the three stages to the library in one call; the
arithmetic is that of ``GnTracker.track`` / ``GnSuppMapper.__call__`` / ``MonoVO.is_kf`` (tests/test_gpu_sequence.py compares the two)."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from ..optim import window as _window
from ..segment_table import table_of

import os as _os
SUPP_CHECK_FIRST = int(_os.environ.get("SP_SUPP_CHECK_FIRST", "1"))
TRACK, SUPP, CRITERION = _lib.SP_CHAIN_TRACK, _lib.SP_CHAIN_SUPP, _lib.SP_CHAIN_CRITERION


def _gn_record(rec, win, level):
    """``SpWindowGn`` of a built ``PoseWindow`` at one pyramid level (what ``PoseWindow.run_gn`` passes ``sp_window_gn_run``)."""
    gn, d = win._gn_state(), win.desc[level]
    rec.pairs, rec.chunks, rec.spans, rec.edges, rec.nodes, rec.blocks = d.data_ptr(), win.chunks.data_ptr(), win.spans.data_ptr(), win.edges.data_ptr(), win.nodes.data_ptr(), win.blocks.data_ptr()
    rec.span_partials, rec.seg_partials, rec.scratch = win.partials.data_ptr(), win.seg_partials.data_ptr(), gn['scratch'].data_ptr()
    rec.nodes_backup, rec.kld_backup, rec.state, rec.losses = gn['nodes_backup'].data_ptr(), gn['kld_backup'].data_ptr(), gn['state'].data_ptr(), gn['losses'].data_ptr()
    rec.n_spans, rec.n_edges, rec.n_nodes, rec.n_blocks = win.n_spans, win.n_edges, win.n_nodes, win.n_sources
    rec.sum_N, rec.max_N, rec.n_unknowns, rec.max_losses = gn['sum_N'], win.max_N, gn['n_y'], win.max_iters


def _bind_window(cw, win, phases, check_first=0):
    """phases: [(pyramid level, max iterations, irls_eps, conv_tol)] -- the schedule ``run_gn`` is called with, phase by phase."""
    assert len(phases) <= _lib.SP_CHAIN_PHASES and max(win.level_ids) < _lib.SP_CHAIN_LEVELS
    for l in range(_lib.SP_CHAIN_LEVELS):
        cw.gn[l].pairs = None
    for l in win.level_ids:
        _gn_record(cw.gn[l], win, l)
    for p, (level, n, eps, tol) in enumerate(phases):
        cw.phase[p].level, cw.phase[p].max_iters, cw.phase[p].irls_eps, cw.phase[p].conv_tol = int(level), int(n), float(eps), float(tol)
    cw.n_phases = len(phases)
    cw.check_every = _window.GN_CHECK_EVERY
    cw.check_first = int(check_first)
    cw.flags = (2 if _window.GN_PREDICTED_EXIT else 0) | (4 if win.depths_fixed else 0)
    cw.lam0, cw.lm_up, cw.lm_down, cw.lm_min = 1e-4, 8.0, 0.5, 1e-7          # (PoseWindow.reset_gn / run_gn defaults)
    cw.state_host = win._gn_state()['state_host'].data_ptr()


class ChainStep:
    """The argument record of ``sp_chain_step`` for one sequence: frame-sized scratch, the per-frame history of tracked poses and
    affine pairs (device), and the windows of the latest keyframe once bound."""

    def __init__(self, n_frames, H, W, n_levels, device):
        self.lib = _lib.load()
        self.device, self.H, self.W = device, int(H), int(W)
        st = self.st = _lib.SpChainStep()
        st.H, st.W, st.n_levels = self.H, self.W, int(n_levels)
        self._levels = []
        h, w = self.H, self.W
        for l in range(1, int(n_levels)):
            h, w = (h + 1) // 2, (w + 1) // 2
            self._levels.append(torch.empty(3, h, w, dtype=torch.float32, device=device))
            st.level[l] = self._levels[-1].data_ptr()
        self.hist_pose = torch.zeros(n_frames, 4, 4, dtype=torch.float32, device=device)     # tracked camera-to-world pose of frame i
        self.hist_aff = torch.zeros(n_frames, 2, dtype=torch.float32, device=device)
        self.keys = torch.empty(self.H * self.W, dtype=torch.int64, device=device)
        self.depth = torch.empty(self.H, self.W, dtype=torch.float32, device=device)
        self.rel_pose = torch.empty(16, dtype=torch.float32, device=device)
        self.crit = torch.empty(4, dtype=torch.float32, device=device)
        self.crit_host = torch.zeros(4, dtype=torch.float32).pin_memory()
        self.crit_ws = torch.zeros(self.lib.sp_kf_criterion_ws_words(), dtype=torch.int32, device=device)
        st.crit_ws = self.crit_ws.data_ptr()
        st.keys, st.depth_out, st.rel_pose, st.crit, st.crit_host = self.keys.data_ptr(), self.depth.data_ptr(), self.rel_pose.data_ptr(), self.crit.data_ptr(), self.crit_host.data_ptr()
        st.valid_thresh = 1e-6
        self.tracker = self.mapper = None
        self._keep = {}

    # ---- the windows of the latest keyframe ---------------------------------------------------------------------------------------------
    def bind_tracker(self, tracker, kf, affine):
        """``tracker``: a built ``loops.GnTracker`` (node 0 = the keyframe, node 1 = the tracked frame)."""
        st, win, sch = self.st, tracker.win, tracker.sch
        phases = [(l, n, sch['irls_eps'], sch['conv_tol']) for l, n in tracker.phases]
        if sch['polish_max'] > 0:
            phases.append((win.level_ids[0], sch['polish_max'], sch['polish_eps'], sch['polish_tol']))
        _bind_window(st.track, win, phases)
        tg = st.track_target
        tg.node = 1
        for l in range(_lib.SP_CHAIN_LEVELS):
            tg.packed[l] = win.trg3[(1, l)].data_ptr() if (1, l) in win.trg3 else None
        assert all(win.level_hw[(1, l)] == ((self.H + (1 << l) - 1) >> l, (self.W + (1 << l) - 1) >> l) for l in win.level_ids)
        # the keyframe the criterion renders: its table, and the depths / pose the TRACKER's window holds (kept equal to the sequence's
        # kf_klds[-1] / kf_poses[-1] by GnTracker.update_keyframe after every mapping)
        table = table_of(kf)
        K = kf.K.detach().float().contiguous().to(self.device)
        st.pix, st.baseL, st.seg_off, st.kp_L = table.pix.data_ptr(), table.baseL.data_ptr(), table.seg_off.data_ptr(), table.kp_L.data_ptr()
        st.kld, st.K, st.kf_pose = win.kld.data_ptr(), K.data_ptr(), win.nodes.data_ptr()      # (SpWindowNode.T is the node's first field)
        st.N, st.P = table.N, table.P
        assert (table.H, table.W) == (self.H, self.W)
        self.affine = bool(affine)
        self.tracker, self.mapper = tracker, None
        self._keep['track'] = (table, K, win)

    def bind_mapper(self, mapper):
        """``mapper``: a built ``loops.GnSuppMapper`` of the same keyframe as the bound tracker."""
        st, win, gn = self.st, mapper.win, mapper.gn
        phases = [(0, min(mapper.num_iters, gn['max_iters']), gn['irls_eps'], gn['conv_tol'])]
        if gn['polish_max'] > 0 and mapper.num_iters > gn['max_iters'] // 2:
            phases.append((0, gn['polish_max'], gn['polish_eps'], gn['polish_tol']))
        _bind_window(st.supp, win, phases, check_first=SUPP_CHECK_FIRST)      # (the depths of a keyframe that has been mapped every frame move in one step)
        for j in range(2):
            tg = st.supp_target[j]
            tg.node = mapper.slots[j]
            for l in range(_lib.SP_CHAIN_LEVELS):
                tg.packed[l] = win.trg3[(mapper.slots[j], l)].data_ptr() if (mapper.slots[j], l) in win.trg3 else None
        assert win.n_sources == 1 and self.tracker is not None and win.Ns[0] == self.tracker.win.Ns[0]
        st.kld_src, st.kld_dst, st.kld_n = win.kld.data_ptr(), self.tracker.win.kld.data_ptr(), int(win.Ns[0])
        self.mapper = mapper

    # ---- one frame ----------------------------------------------------------------------------------------------------------------------
    def run(self, stages, i=None, image=None, start_pose=None, start_aff=None, prev=None, supp_images=0, supp_one=False, pose=None):
        """``stages``: TRACK | SUPP | CRITERION.  TRACK: frame ``i`` with planar ``image`` from ``start_pose`` (/ ``start_aff``); the result
        lands in ``hist_pose[i]`` / ``hist_aff[i]``.  SUPP: ``prev`` = history index of the older running supporting frame (the newer one
        is ``i``; ``supp_one``: there is only frame ``i`` -- ``prev`` = ``i``, the first slot's edge carries no weight).  CRITERION without TRACK: ``pose`` = the (4,4) device tensor to judge.  Returns (tracker iterations, mapper iterations,
        criterion [validity ratio, scale, translation difference, rotation degrees] | None)."""
        st = self.st
        keep = []
        st.stages = int(stages)
        if stages & TRACK:
            img = image[:3].detach().float().contiguous()
            assert img.shape[-2:] == (self.H, self.W) and img.device == self.hist_pose.device
            sp = start_pose.detach().float().contiguous()
            assert sp.device == self.hist_pose.device and sp.numel() == 16, "the start pose lives on the chain's device"
            keep += [img, sp]
            st.image = img.data_ptr()
            st.track_target.pose = sp.data_ptr()
            if self.affine:
                sa = start_aff.detach().float().contiguous()
                assert sa.device == self.hist_pose.device and sa.numel() == 2
                keep.append(sa)
                st.track_target.aff = sa.data_ptr()
                st.out_aff = self.hist_aff[i].data_ptr()
            else:
                st.track_target.aff, st.out_aff = None, None
            st.out_pose = self.hist_pose[i].data_ptr()
        elif stages & CRITERION:
            p = pose.detach().float().contiguous()
            assert p.device == self.hist_pose.device and p.numel() == 16
            keep.append(p)
            st.out_pose = p.data_ptr()
        if stages & SUPP:
            assert self.mapper is not None
            for j, idx in enumerate((prev, i)):
                st.supp_target[j].pose = self.hist_pose[idx].data_ptr()
                st.supp_target[j].aff = self.hist_aff[idx].data_ptr() if self.affine else None
            st.supp_images = int(supp_images)
            edges = (self.mapper.edges_one if supp_one else self.mapper.edges_two).data_ptr()     # (one running frame: slot 0's edge off)
            for l in self.mapper.win.level_ids:
                st.supp.gn[l].edges = edges
        rc = self.lib.sp_chain_step(ctypes.byref(st), _lib.stream_ptr())
        _lib.check(rc, "sp_chain_step")
        for w in ((self.tracker.win,) if stages & TRACK else ()) + ((self.mapper.win,) if stages & SUPP else ()):
            w._gn.pop('host_stale', None)                # (the call left the pinned copy of the LM state current)
            w._gn['host_seen'] = True
        crit = self.crit_host.tolist() if stages & CRITERION else None
        return st.track_iters, st.supp_iters, crit
