"""Fused optimiser of the reference's Adam loops: poses, per-keyframe log-depths and affine
brightness pairs of a small WINDOW of frames, optimised entirely on the GPU.

The reference runs three loop shapes around its cost functions -- two-frame SfM (``odometery/two_frame_sfm.py:116-207``),
frame-to-keyframe tracking (``odometery/odometery.py:300-312,375-407``) and windowed mapping (``:576-648,756-915``) -- each
as eager PyTorch: autograd through ``lietorch`` + ``torch.optim.Adam`` + a dozen small pose-algebra launches per
parameter (``torch.linalg.inv``, ``renormalise_se3``, tangent zeroing).  Behind a fused cost kernel that glue is >95 % of an
iteration.  Here one iteration of ANY of the three is

    sp_pairs_cost (mode 0)      every (source keyframe -> target frame) edge of the window in ONE launch
    sp_window_step              per-edge reduction; chain rule onto the pose tangents; Adam with torch semantics; fold-in
                                T <- T inv(Exp(d)); renormalise; tangent reset; relative-loss early stop; next relative
                                poses -- two launches, no host synchronisation, capturable in a hipGraph.

``PoseWindow`` is the generic object; ``odometery/two_frame_sfm.py``, ``odometery/loops.py`` build it for their loop.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib
from ..segment_table import _Ident, _same, packed_target, table_of
from .batch_prepare import flat_work_list, stage
from .pair_batch import DEFAULT_BATCH_TILE_POINTS, DEFAULT_SPAN_POINTS, GRANULE, MIN_SPANS, _level_images, pad_layout, pad_points

KIND_WINDOW, KIND_DIRECT = 0, 1
_SP_PAIR_DT = np.dtype(_lib.SpPair)


def _upload_struct_array(arr, device):
    return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)


# How often the host loop of a Gauss-Newton phase looks at the device's state (PoseWindow.run_gn).  Launches behind the converged
# iteration change nothing but still RUN -- the cost pass does not know the window is frozen -- so on the few-edge windows of the
# config-3 chain a poll per iteration (a 64-byte copy and a stream synchronisation, ~15 us) is cheaper than three wasted iterations of
# ~100 us; SP_GN_CHECK_EVERY overrides (tools/chain_profile.py measures).
import os as _os
GN_CHECK_EVERY = int(_os.environ.get("SP_GN_CHECK_EVERY", "4"))
GN_PREDICTED_EXIT = _os.environ.get("SP_GN_PREDICTED_EXIT", "1") != "0"


class PoseWindow:
    def __init__(self, sources, nodes, edges, levels, abs_loss=False, skip_first=False, rel_tol=0.0, use_affine=False,
                 max_iters=4096, tile_points=DEFAULT_BATCH_TILE_POINTS, span_points=None, private_targets=None,
                 share_sources=False):
        """sources: list of dict(kf=KeyFrame, kld=(N,) tensor, lr=float [0 = frozen], node=int [-1 = identity pose]);
        nodes:   list of dict(T=(4,4), kind=KIND_WINDOW|KIND_DIRECT, lr_pose=float, lr_aff=float, aff=(2,)|None,
                              renorm=bool, image=(3,H,W)|None, K=(3,3)|None)  -- image/K needed when the node is a target;
        edges:   list of (source index, target node, weight, zmin);
        levels:  (pyramid_min, pyramid_max) like ``config['aligment']`` (max exclusive);
        private_targets: target nodes whose images this window REPLACES in place (``set_target_image`` / ``copy_target_image`` / the slots of
                 ``sp_chain_step``): they get buffers of their own.  None = all of them.  Every other target node at pyramid level 0 reads
                 the packed image cached on the frame's tensor (``segment_table.packed_target``), shared by all windows the frame is part of;
        share_sources: keep / re-use the padded source samples of every source keyframe with its segment table (see below; the Gauss-Newton
                 windows of the odometry chain -- a keyframe is the source of a dozen windows while it lives).  Off, every window samples
                 its own under the depths it is built with, as the reference does per iteration."""
        lib = _lib.load()
        self.lib = lib
        dev = sources[0]['kf'].image.device
        _lib.require_device(sources[0]['kf'].image)
        self.device = dev
        self.level_ids = list(range(levels[0], levels[1]))
        max_level = levels[1] - 1
        self.abs_loss, self.skip_first, self.rel_tol = int(abs_loss), int(skip_first), float(rel_tol)
        self.n_sources, self.n_nodes, self.n_edges = len(sources), len(nodes), len(edges)
        S, E = self.n_sources, self.n_edges
        # ---- source keyframes: padded segment tables, per-level source samples --------------------------------
        tables = [table_of(s['kf']) for s in sources]
        self.Ns = [t.N for t in tables]
        self.max_N = max(self.Ns)
        # (what of this is a property of the KEYFRAME is kept with its table: the padded layout, the padded points, and the padded source
        #  samples per level.  A keyframe is the source of several windows while it lives -- its tracker, its supplementary-mapping window,
        #  every scheduled mapping it takes part in -- and the samples do not depend on the depths beyond the last bit of the re-projected
        #  pixel (core/dense_optim.py:143-162; the tolerance the persistent windows already carry, tests/test_gpu_sequence.py))
        pads, self.src4, self.pix = [], {}, []
        for k, (s, tab) in enumerate(zip(sources, tables)):
            kfk = s['kf']
            wc = getattr(tab, '_window_cache', None) if share_sources else None
            if wc is None or not _same(wc['idents'], (kfk.image, kfk.K)):
                wc = dict(idents=(_Ident(kfk.image), _Ident(kfk.K)), pads=pad_layout(tab.counts, dev))
                if share_sources:
                    tab._window_cache = wc
            pads.append(wc['pads'])
            lv = None
            for l in self.level_ids:
                hit = wc.get(l)
                if hit is None:
                    if lv is None:
                        lv = _level_images(kfk.image[:3].float(), max_level)
                    hit = wc[l] = pad_points(tab.source_level(lv[l], kfk.K, s['kld'].to(dev), cache=False).reshape(-1, 4), wc['pads']).reshape(-1)
                self.src4[(k, l)] = hit
            if 'pix' not in wc:
                wc['pix'] = pad_points(tab.pix, wc['pads'])                         # after source_level(): validity bits set
            self.pix.append(wc['pix'])
        self.kp_L = [t.kp_L for t in tables]
        # ---- log-depth blocks ----------------------------------------------------------------------------------
        n_off = np.concatenate(([0], np.cumsum(self.Ns)))
        self.n_off = n_off
        self.kld = torch.cat([s['kld'].detach().float().to(dev).reshape(-1) for s in sources]).contiguous()
        self.kld_m = torch.zeros_like(self.kld)
        self.kld_v = torch.zeros_like(self.kld)
        blocks = (_lib.SpWindowBlock * S)()
        for k, s in enumerate(sources):
            blocks[k].kld = self.kld.data_ptr() + 4 * int(n_off[k])
            blocks[k].m = self.kld_m.data_ptr() + 4 * int(n_off[k])
            blocks[k].v = self.kld_v.data_ptr() + 4 * int(n_off[k])
            blocks[k].N = self.Ns[k]
            blocks[k].lr = float(s.get('lr', 0.0))
        self._blocks_host = blocks
        self.depths_fixed = all(float(s_.get('lr', 0.0)) == 0.0 for s_ in sources)     # (sp_window_gn_step flags bit 2: no Schur launch)
        # ---- pose nodes ----------------------------------------------------------------------------------------
        arr = (_lib.SpWindowNode * self.n_nodes)()
        # (ONE read-back for all poses and one for all affine pairs: a .cpu() per node was a host synchronisation each)
        Ts = torch.stack([nd['T'].detach().float().reshape(4, 4).to(dev) for nd in nodes]).cpu().numpy().reshape(-1, 16)
        with_aff = [i for i, nd in enumerate(nodes) if nd.get('aff') is not None]
        affs = torch.stack([nodes[i]['aff'].detach().float().reshape(2).to(dev) for i in with_aff]).cpu().numpy() if with_aff else None
        for i, nd in enumerate(nodes):
            arr[i].T = (ctypes.c_float * 16)(*Ts[i])
            arr[i].lr_pose = float(nd.get('lr_pose', 0.0))
            arr[i].lr_aff = float(nd.get('lr_aff', 0.0))
            arr[i].kind = int(nd.get('kind', KIND_WINDOW))
            arr[i].flags = 1 if nd.get('renorm', False) else 0
        for j, i in enumerate(with_aff):
            arr[i].aff = (ctypes.c_float * 2)(*affs[j])
        nodes_host = arr
        self._all_kind0 = all(int(nd.get('kind', KIND_WINDOW)) == KIND_WINDOW for nd in nodes)
        # ---- target images: packed per level -------------------------------------------------------------------
        self.trg3, self.level_hw = {}, {}
        targets = sorted({e[1] for e in edges})
        self._shared_targets = set()
        for i in targets:
            img = nodes[i]['image']
            assert img is not None and nodes[i].get('K') is not None, f"node {i} is a target but has no image / K"
            if private_targets is not None and i not in private_targets and self.level_ids == [0] and img.is_cuda and img.dim() == 3:
                self.trg3[(i, 0)] = packed_target(img).reshape(-1)
                self.level_hw[(i, 0)] = tuple(img.shape[-2:])
                self._shared_targets.add(i)
                continue
            lv = _level_images(img[:3].float().to(dev), max_level)
            for l in self.level_ids:
                Hl, Wl = lv[l].shape[-2:]
                packed = torch.empty(Hl * Wl * 3, dtype=torch.float32, device=dev)
                _lib.check(lib.sp_pack_rgb(_lib.ptr(lv[l].contiguous()), 1, Hl, Wl, _lib.ptr(packed), _lib.stream_ptr()), "sp_pack_rgb")
                self.trg3[(i, l)] = packed
                self.level_hw[(i, l)] = (Hl, Wl)
        # ---- edges = pairs of the many-pairs cost path ---------------------------------------------------------
        earr = (_lib.SpWindowEdge * E)()
        for e, (k, i, wgt, _z) in enumerate(edges):
            assert 0 <= k < S and 0 <= i < self.n_nodes
            assert nodes[i].get('kind', KIND_WINDOW) == KIND_WINDOW or sources[k].get('node', -1) < 0
            earr[e].src_node, earr[e].trg_node, earr[e].block, earr[e].weight = int(sources[k].get('node', -1)), i, k, float(wgt)
        self.edge_list = [(int(k), int(i), float(wgt), float(z)) for k, i, wgt, z in edges]
        epads = [pads[k] for k, _, _, _ in edges]
        total = sum(pd['Ppad'] for pd in epads)
        if span_points is None:
            span_points = min(DEFAULT_SPAN_POINTS, total // MIN_SPANS)
        self.span_points = max(int(span_points), GRANULE)
        # the work list of all edges at once by the library's host helper (the Python form took 1.8 ms for a reference-sized window)
        e_N = np.array([len(pd['pc']) for pd in epads], dtype=np.int64)
        wl = flat_work_list(np.concatenate([pd['pc'] for pd in epads]), np.concatenate([pd['pseg_off'][:-1] for pd in epads]),
                            np.concatenate(([0], np.cumsum(e_N))), self.span_points, tile_points, GRANULE)
        self.n_chunks, self.n_spans = len(wl['chunks']), len(wl['spans'])
        sto_off = wl['sto_off']
        # (every small host array of the window goes to the device through ONE pinned buffer and one copy at the end: a copy from pageable
        #  memory each -- a dozen per window -- made the host wait for the stream every time)
        self.seg_tile_off = torch.empty(max(len(wl['seg_tile_off']), 1), dtype=torch.int32, device=dev)
        self.pose_slots = torch.zeros(E, 16, dtype=torch.float32, device=dev)
        self.aff_slots = torch.zeros(E, 4, dtype=torch.float32, device=dev) if use_affine else None
        self.Ps = [tables[k].P for k, _, _, _ in edges]
        self.desc = {}
        Ks_src = torch.stack([s_['kf'].K.detach().float().to(dev) for s_ in sources]).cpu().numpy()
        Ks_trg = dict(zip(targets, torch.stack([nodes[i]['K'].detach().float().to(dev) for i in targets]).cpu().numpy()))
        # (numpy records, one column at a time: a ctypes assignment per field, edge and level took 1.3 ms of a reference-sized window's 4)
        e_src = np.array([k for k, _, _, _ in edges], dtype=np.int64)
        e_trg = np.array([i for _, i, _, _ in edges], dtype=np.int64)
        base = np.zeros(E, dtype=_SP_PAIR_DT)
        base['pix'] = np.array([t.data_ptr() for t in self.pix], dtype=np.uint64)[e_src]
        base['kp_L'] = np.array([t.data_ptr() for t in self.kp_L], dtype=np.uint64)[e_src]
        base['kld'] = (self.kld.data_ptr() + 4 * n_off[e_src].astype(np.int64)).astype(np.uint64)
        base['pose'] = (self.pose_slots.data_ptr() + 64 * np.arange(E, dtype=np.int64)).astype(np.uint64)
        if use_affine:
            base['aff'] = (self.aff_slots.data_ptr() + 16 * np.arange(E, dtype=np.int64)).astype(np.uint64)
        base['seg_tile_off'] = (self.seg_tile_off.data_ptr() + 4 * np.asarray(sto_off, dtype=np.int64)[:E]).astype(np.uint64)
        Kt_all = np.stack([Ks_trg[i] for i in targets])
        t_pos = {i: q for q, i in enumerate(targets)}
        e_tpos = np.array([t_pos[int(i)] for i in e_trg], dtype=np.int64)
        base['K_src'] = np.stack((Ks_src[:, 0, 0], Ks_src[:, 1, 1], Ks_src[:, 0, 2], Ks_src[:, 1, 2]), axis=1)[e_src]
        base['K_trg'] = np.stack((Kt_all[:, 0, 0], Kt_all[:, 1, 1], Kt_all[:, 0, 2], Kt_all[:, 1, 2]), axis=1)[e_tpos]
        for name, vals in (('N', [t.N for t in tables]), ('P', [t.P for t in tables]), ('H', [t.H for t in tables]), ('W', [t.W for t in tables])):
            base[name] = np.array(vals, dtype=np.int32)[e_src]
        s_off, c_off = np.asarray(wl['s_off'], dtype=np.int64), np.asarray(wl['c_off'], dtype=np.int64)
        base['tile0'], base['n_tiles'] = s_off[:E], s_off[1:E + 1] - s_off[:E]
        base['zmin'] = np.array([z for _, _, _, z in edges], dtype=np.float32)
        base['rec0'] = 4 * c_off[:E]
        desc_host = {}
        for l in self.level_ids:
            parr = base.copy()
            parr['src4'] = np.array([self.src4[(k, l)].data_ptr() for k in range(S)], dtype=np.uint64)[e_src]
            parr['trg3'] = np.array([self.trg3[(i, l)].data_ptr() for i in targets], dtype=np.uint64)[e_tpos]
            hw_l = np.array([self.level_hw[(i, l)] for i in targets], dtype=np.int32)[e_tpos]
            parr['Hl'], parr['Wl'] = hw_l[:, 0], hw_l[:, 1]
            desc_host[l] = parr.view(np.uint8).reshape(-1)
        # ---- workspaces / optimiser state ----------------------------------------------------------------------
        # (sized for the Gauss-Newton records, the larger of the two optimisers')
        self.partials = torch.empty(self.n_spans * _lib.SP_GNA_PARTIAL_FLOATS, dtype=torch.float32, device=dev)
        self.seg_partials = torch.empty(4 * self.n_chunks * _lib.SP_GNA_SEG_FLOATS, dtype=torch.float32, device=dev)
        self._gn = None
        self.scratch = torch.zeros(lib.sp_window_scratch_doubles(E, self.max_N), dtype=torch.float64, device=dev)
        self.state = torch.zeros(12, dtype=torch.float32, device=dev)
        self.max_iters = int(max_iters)
        self.loss_hist = torch.zeros(self.max_iters, dtype=torch.float32, device=dev)
        self._graphs = {}
        as_bytes = lambda a: np.frombuffer(bytes(a), dtype=np.uint8)
        up = stage([as_bytes(blocks), as_bytes(nodes_host), as_bytes(earr), np.ascontiguousarray(wl['chunks'], dtype=np.int32),
                    np.ascontiguousarray(wl['spans'], dtype=np.int32), np.ascontiguousarray(wl['seg_tile_off'], dtype=np.int32)]
                   + [desc_host[l] for l in self.level_ids], dev)
        self.blocks, self.nodes, self.edges, self.chunks, self.spans = up[:5]
        self.seg_tile_off[: up[5].numel()].copy_(up[5])          # (the descriptors hold addresses inside this array: it was allocated above)
        for l, d in zip(self.level_ids, up[6:]):
            self.desc[l] = d
        self._keep = (tables, pads, up)
        self.compose()

    # ----------------------------------------------------------------------------------------------------------
    def compose(self):
        l = self.level_ids[0]
        _lib.check(self.lib.sp_window_compose(_lib.ptr(self.desc[l]), _lib.ptr(self.edges), self.n_edges, _lib.ptr(self.nodes),
                                              self.n_nodes, _lib.stream_ptr()), "sp_window_compose")

    def step(self, level):
        """One Adam iteration of the whole window at pyramid ``level`` (3 launches, nothing returns to the host)."""
        d = self.desc[level]
        _lib.check(self.lib.sp_pairs_cost(_lib.ptr(d), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, 0, 0.0,
                                          _lib.ptr(self.partials), _lib.ptr(self.seg_partials), _lib.stream_ptr()), "sp_pairs_cost")
        _lib.check(self.lib.sp_window_step(_lib.ptr(d), _lib.ptr(self.edges), self.n_edges, _lib.ptr(self.nodes), self.n_nodes,
                                           _lib.ptr(self.blocks), self.n_sources, self.max_N, _lib.ptr(self.partials),
                                           _lib.ptr(self.seg_partials), _lib.ptr(self.scratch), self.abs_loss, self.skip_first,
                                           self.rel_tol, _lib.ptr(self.state), _lib.ptr(self.loss_hist), self.max_iters,
                                           _lib.stream_ptr()), "sp_window_step")

    # ---- re-use of a built window for new data (same graph, same source keyframes) --------------------------------------
    def set_target_image(self, node, image):
        """Replace the image of target node ``node`` IN PLACE (same size, same intrinsics): its pyramid levels are rebuilt and
        packed into the buffers the edge descriptors already point to -- a tracker keeps ONE window per keyframe and feeds it the
        frames (the source tables, samples, work list and descriptors are a per-keyframe cost, not a per-frame one)."""
        assert node not in self._shared_targets, "set_target_image: the node reads a shared image (name it in private_targets)"
        lv = _level_images(image[:3].float().to(self.device), max(self.level_ids))
        for l in self.level_ids:
            Hl, Wl = lv[l].shape[-2:]
            assert (Hl, Wl) == self.level_hw[(node, l)], "set_target_image: the frame size changed; build a new window"
            _lib.check(self.lib.sp_pack_rgb(_lib.ptr(lv[l].contiguous()), 1, Hl, Wl, _lib.ptr(self.trg3[(node, l)]), _lib.stream_ptr()), "sp_pack_rgb")

    def copy_target_image(self, dst_node, src_node):
        """The packed pyramid of target node ``src_node`` into ``dst_node``'s buffers (same size): a frame that moves from one slot of
        a persistent window to another is not packed twice."""
        assert dst_node not in self._shared_targets
        for l in self.level_ids:
            assert self.level_hw[(dst_node, l)] == self.level_hw[(src_node, l)]
            self.trg3[(dst_node, l)].copy_(self.trg3[(src_node, l)])

    def _nodes_f32(self):
        """The node array as (n_nodes, 44) floats ON THE DEVICE (SpWindowNode: T 0..15, a 16..21, m 22..27, v 28..33, aff 34..35, aff_m 36..37,
        aff_v 38..39, lr_pose 40, lr_aff 41, kind / flags 42..43 as raw bits)."""
        return self.nodes.view(torch.float32).view(self.n_nodes, 44)

    def set_nodes(self, updates):
        """updates: {node index: dict(T=(4,4) [, aff=(2,)])}: overwrite poses / affine pairs (tangents and Adam moments cleared), then
        re-compose every edge's relative pose.  On the device: the node array is neither read back nor uploaded (round 6; the host round
        trip -- a synchronisation per pose -- was a third of a scheduled mapping's bookkeeping in the config-3 chain)."""
        if not updates:
            return
        nf = self._nodes_f32()
        ids = list(updates)
        idx = torch.tensor(ids, dtype=torch.long).to(self.device, non_blocking=True)
        nf[idx, :16] = torch.stack([updates[i]['T'].detach().to(self.device, torch.float32).reshape(16) for i in ids])
        nf[idx, 16:34] = 0.0
        nf[idx, 36:40] = 0.0
        with_aff = [i for i in ids if updates[i].get('aff') is not None]
        if with_aff:
            ia = idx if len(with_aff) == len(ids) else torch.tensor(with_aff, dtype=torch.long).to(self.device, non_blocking=True)
            nf[ia, 34:36] = torch.stack([updates[i]['aff'].detach().to(self.device, torch.float32).reshape(2) for i in with_aff])
        self.compose()

    def set_klds(self, klds):
        """Overwrite the log-depth blocks (one (N_k,) tensor per source keyframe); Adam moments cleared."""
        self.kld.copy_(torch.cat([k.detach().float().to(self.device).reshape(-1) for k in klds]))
        self.kld_m.zero_(); self.kld_v.zero_()

    # ---- Gauss-Newton / LM (sp_window_gn_step) ---------------------------------------------------------------------
    def _gn_state(self):
        if self._gn is None:
            sum_N = int(self.n_off[-1])
            n_y = 0
            for nd in self._node_array():
                n_y += (6 if nd.lr_pose > 0 else 0) + (2 if nd.lr_aff > 0 else 0)
            if n_y > 512 or self.n_nodes > 64 or self.n_sources > 64:
                raise ValueError(f"{n_y} camera unknowns / {self.n_nodes} nodes / {self.n_sources} source keyframes exceed the 512 / 64 / 64 "
                                 "the Gauss-Newton window solver holds; use the Adam optimiser")
            self._gn = dict(
                scratch=torch.zeros(self.lib.sp_window_gn_scratch_doubles(self.n_edges, self.n_sources, sum_N, self.max_N, n_y), dtype=torch.float64, device=self.device),
                nodes_backup=torch.zeros_like(self.nodes), kld_backup=torch.zeros(sum_N, dtype=torch.float32, device=self.device),
                state=torch.zeros(16, dtype=torch.float32, device=self.device), state_host=torch.zeros(16, dtype=torch.float32).pin_memory(),
                losses=torch.zeros(self.max_iters, dtype=torch.float32, device=self.device), sum_N=sum_N, n_y=n_y,
                phase_keep=torch.tensor([1, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1], dtype=torch.float32, device=self.device),
                phase_set=torch.tensor([0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=self.device))
            self.reset_gn()
        return self._gn

    def reset_gn(self, lam=1e-4):
        """Fresh LM state (lambda, no accepted point yet, not converged); the loss history restarts."""
        gn = self._gn_state() if self._gn is None else self._gn
        gn['state'].zero_()
        gn['state'][0] = lam
        gn['state'][1] = -1.0
        gn['state_host'].zero_()
        gn['host_seen'] = False

    def begin_gn_phase(self):
        """A new phase of a schedule (another pyramid level / IRLS epsilon): losses of different phases are not comparable, so
        the accept test and the convergence test start afresh; lambda and the iteration count carry over."""
        gn = self._gn_state()
        gn['state'].mul_(gn['phase_keep']).add_(gn['phase_set'])          # [1] = -1, [4] = [6] = 0 in two launches, no host value involved
        gn['host_seen'] = False

    def gn_step(self, level, irls_eps=1e-3, pose_only=False, conv_tol=0.0, lm_up=8.0, lm_down=0.5, lm_min=1e-7):
        """One Gauss-Newton / LM iteration of the whole window at pyramid ``level`` (3 launches, nothing returns to the host):
        the cost pass in mode 2 (normal equations incl. the affine brightness columns) + ``sp_window_gn_step``."""
        gn = self._gn_state()
        d = self.desc[level]
        _lib.check(self.lib.sp_pairs_cost(_lib.ptr(d), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, 2, float(irls_eps),
                                          _lib.ptr(self.partials), _lib.ptr(self.seg_partials), _lib.stream_ptr()), "sp_pairs_cost")
        _lib.check(self.lib.sp_window_gn_step(_lib.ptr(d), _lib.ptr(self.edges), self.n_edges, _lib.ptr(self.nodes), self.n_nodes,
                                              _lib.ptr(self.blocks), self.n_sources, gn['sum_N'], self.max_N, gn['n_y'], _lib.ptr(self.partials),
                                              _lib.ptr(self.seg_partials), _lib.ptr(gn['scratch']), _lib.ptr(gn['nodes_backup']),
                                              _lib.ptr(gn['kld_backup']), (1 if pose_only else 0) | (4 if self.depths_fixed else 0), float(lm_up), float(lm_down), float(lm_min),
                                              float(conv_tol), _lib.ptr(gn['state']), _lib.ptr(gn['losses']), self.max_iters,
                                              _lib.stream_ptr()), "sp_window_gn_step")
        gn['host_stale'] = True

    def run_gn(self, level, max_iters, irls_eps=1e-3, conv_tol=2e-3, pose_only=False, check_every=None, lm_up=8.0, lm_down=0.5, lm_min=1e-7,
               predicted_exit=None):
        """Up to ``max_iters`` LM iterations at ``level`` as ONE phase: stops once an accepted step lowers the loss by less than
        ``conv_tol`` of it (the device freezes the window -- launches after that change nothing; the host loop, ONE foreign call
        ``sp_window_gn_run``, looks at the state every ``check_every`` iterations).  Returns the iterations the phase really took
        (evaluations of the cost, rejected ones and the final converged-test evaluation included), from the device's counter.
        ``predicted_exit`` (default GN_PREDICTED_EXIT): the phase also ends right after a step PREDICTED to buy less than ``conv_tol`` of the
        loss (flags bit 1 of sp_window_gn_step; the pair solver's SP_PHASE_PREDICTED_EXIT) -- without the evaluation that confirms it."""
        gn = self._gn_state()
        if check_every is None:
            check_every = GN_CHECK_EVERY
        if predicted_exit is None:
            predicted_exit = GN_PREDICTED_EXIT
        self.begin_gn_phase()
        if gn.pop('host_stale', False):             # (gn_step() moved the device's counter since the host last saw it)
            gn['state_host'].copy_(gn['state'])
        n0 = int(gn['state_host'][5])
        d = self.desc[level]
        rc = self.lib.sp_window_gn_run(_lib.ptr(d), _lib.ptr(self.chunks), _lib.ptr(self.spans), self.n_spans, float(irls_eps), _lib.ptr(self.edges),
                                       self.n_edges, _lib.ptr(self.nodes), self.n_nodes, _lib.ptr(self.blocks), self.n_sources, gn['sum_N'],
                                       self.max_N, gn['n_y'], _lib.ptr(self.partials), _lib.ptr(self.seg_partials), _lib.ptr(gn['scratch']),
                                       _lib.ptr(gn['nodes_backup']), _lib.ptr(gn['kld_backup']), (1 if pose_only else 0) | (2 if predicted_exit else 0) | (4 if self.depths_fixed else 0), float(lm_up),
                                       float(lm_down), float(lm_min), float(conv_tol), _lib.ptr(gn['state']), _lib.ptr(gn['losses']),
                                       self.max_iters, int(max_iters), int(check_every), gn['state_host'].data_ptr(), _lib.stream_ptr())
        if rc < 0:
            _lib.check(rc, "sp_window_gn_run")
        gn['host_seen'] = int(max_iters) > 0            # (the run's last poll left the pinned copy current)
        return int(gn['state_host'][5]) - n0

    def _gn_host_state(self):
        """The 16-float LM state on the host: the pinned copy ``run_gn`` / ``sp_chain_step`` left behind when nothing has moved the device's
        since (no read-back, no synchronisation), else a fresh copy."""
        gn = self._gn_state()
        if gn.get('host_stale', False) or not gn.get('host_seen', False):
            gn['state_host'].copy_(gn['state'])
            gn.pop('host_stale', None)
            gn['host_seen'] = True
        return gn['state_host']

    def gn_converged(self):
        return bool(float(self._gn_host_state()[6]) != 0)

    def gn_iterations(self):
        return int(self._gn_host_state()[5])

    def gn_losses(self):
        gn = self._gn_state()
        return gn['losses'][: min(self.gn_iterations(), self.max_iters)].clone()

    def gn_profile(self):
        """Microseconds the LAST update kernel spent in its phases (diagnostics): dict of phase -> us."""
        gn = self._gn_state()
        off = self.lib.sp_window_gn_profile_offset(self.n_edges, self.n_sources, gn['sum_N'], self.max_N, gn['n_y'])
        t = gn['scratch'][off: off + 16].cpu().numpy()
        names = ("decision", "backup+clear", "assembly", "schur terms", "factorisation", "substitutions", "depth steps", "poses", "compose")
        out = {n: float(t[i + 1] - t[i]) / 100.0 for i, n in enumerate(names)}
        if t[15] > t[10] > 0:          # one block step of the factorisation (the second), phase by phase
            for i, n in enumerate(("panel out", "diagonal block", "panel rows", "barrier", "trailing update")):
                out["step:" + n] = float(t[11 + i] - t[10 + i]) / 100.0
        return out

    def gn_stats(self):
        st = self._gn_host_state().numpy()
        return dict(lam=float(st[0]), accepted=int(st[2]), rejected=int(st[3]), iterations=int(st[5]), converged=bool(st[6]), failed_solves=int(st[8]),
                    too_many_unknowns=bool(st[9]))

    def run(self, level, iters, graph_chunk=25):
        """``iters`` iterations at ``level``.  ``graph_chunk`` > 0: iterations are replayed from a hipGraph holding that
        many (launch overhead off the critical path); with an early-stop tolerance the converged flag is polled between
        replays -- a frozen window ignores the remaining launches, so results do not depend on the chunking."""
        done = 0
        if graph_chunk and iters >= 2 * graph_chunk:
            g = self._graphs.get((level, graph_chunk))
            if g is None:
                self.step(level)                      # warm-up outside capture (module load); it is a real iteration
                done += 1
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(graph_chunk):
                        self.step(level)
                self._graphs[(level, graph_chunk)] = g
            while iters - done >= graph_chunk:
                g.replay()
                done += graph_chunk
                if self.rel_tol > 0 and self.converged():
                    return
        for _ in range(iters - done):
            self.step(level)

    def set_tangents(self, a):
        """Overwrite the (n_nodes, 6) tangents (resuming a persistent-tangent run)."""
        arr = self._node_array()
        a = a.detach().float().cpu().numpy()
        for i, nd in enumerate(arr):
            nd.a = (ctypes.c_float * 6)(*a[i])
        self.nodes.copy_(_upload_struct_array(arr, self.device))
        self.compose()

    def reset_optimiser(self, lr_scale=None):
        """A fresh Adam (zero moments and step count), like constructing a new torch optimiser over the same parameters."""
        arr = self._node_array()
        for nd in arr:
            nd.m = (ctypes.c_float * 6)(); nd.v = (ctypes.c_float * 6)()
            nd.aff_m = (ctypes.c_float * 2)(); nd.aff_v = (ctypes.c_float * 2)()
            if lr_scale is not None:
                nd.lr_pose *= lr_scale; nd.lr_aff *= lr_scale
        self.nodes.copy_(_upload_struct_array(arr, self.device))
        if lr_scale is not None:
            for bk in self._blocks_host:
                bk.lr *= lr_scale
            self.blocks.copy_(_upload_struct_array(self._blocks_host, self.device))
        self.kld_m.zero_(); self.kld_v.zero_()
        self.state[0] = 0.0

    # ---- results -----------------------------------------------------------------------------------------------
    def converged(self):
        return bool(self.state[3].item() != 0)

    def iterations(self):
        return int(self.state[1].item())

    def losses(self):
        return self.loss_hist[: min(self.iterations(), self.max_iters)].clone()

    def _node_array(self):
        raw = self.nodes.cpu().numpy().tobytes()
        return (_lib.SpWindowNode * self.n_nodes).from_buffer_copy(raw)

    def node_poses(self):
        """(n_nodes,4,4): kind 0 -> the folded-in pose T; kind 1 -> Exp(a) X."""
        if self._all_kind0:                      # (tracking / mapping windows: the poses are a slice of the node array, no read-back)
            return self._nodes_f32()[:, :16].reshape(self.n_nodes, 4, 4).clone()
        arr = self._node_array()
        out = []
        for nd in arr:
            T = torch.tensor(list(nd.T), dtype=torch.float32).reshape(4, 4)
            if nd.kind == KIND_DIRECT:
                from ..lie.se3 import se3_exp_matrix
                T = (se3_exp_matrix(torch.tensor(list(nd.a), dtype=torch.float64)[None])[0] @ T.double()).float()
            out.append(T)
        return torch.stack(out).to(self.device)

    def node_tangents(self):
        return torch.tensor([list(nd.a) for nd in self._node_array()], dtype=torch.float32, device=self.device)

    def node_affines(self):
        return self._nodes_f32()[:, 34:36].clone()

    def klds(self):
        return [self.kld[self.n_off[k]: self.n_off[k + 1]].clone() for k in range(self.n_sources)]

    def edge_poses(self):
        return self.pose_slots.reshape(-1, 4, 4).clone()


class PoseWindowBatch:
    """S independent windows optimised SIDE BY SIDE by Gauss-Newton (config 3's throughput form -- S sequences at once -- as ``PairBatch`` is
    config 2's; include/sp_hip.h ``sp_window_gn_run_multi``).  One window's iteration is four small dependent launches (cost pass over a
    few dozen edges, per-edge reduce, per-keyframe Schur terms, one update workgroup) that leave the chip empty; here every launch covers
    ALL the windows (window = blockIdx.z; the cost pass walks the S work lists in one grid) and one host loop polls all states with one
    copy.  ``windows``: built ``PoseWindow`` objects (any mix of sizes below 192 camera unknowns, or all above).  Per window the arithmetic
    is that of ``PoseWindow.run_gn`` -- bitwise (tests/test_gpu_window_gn.py)."""

    def __init__(self, windows):
        assert len(windows) >= 1
        self.windows = list(windows)
        self.lib = windows[0].lib
        self.device = windows[0].device
        S = len(self.windows)
        for w in self.windows:
            w._gn_state()
        self.args_dev = torch.empty(S * self.lib.sp_window_gn_multi_bytes() + 64, dtype=torch.uint8, device=self.device)
        self.states_dev = torch.zeros(S * 16, dtype=torch.float32, device=self.device)
        self.states_host = torch.zeros(S * 16, dtype=torch.float32).pin_memory()
        self._desc = {}

    def _records(self, level):
        if level not in self._desc:
            arr = (_lib.SpWindowGn * len(self.windows))()
            for i, w in enumerate(self.windows):
                gn, d, a = w._gn_state(), w.desc[level], arr[i]
                a.pairs, a.chunks, a.spans, a.edges, a.nodes, a.blocks = d.data_ptr(), w.chunks.data_ptr(), w.spans.data_ptr(), w.edges.data_ptr(), w.nodes.data_ptr(), w.blocks.data_ptr()
                a.span_partials, a.seg_partials, a.scratch = w.partials.data_ptr(), w.seg_partials.data_ptr(), gn['scratch'].data_ptr()
                a.nodes_backup, a.kld_backup, a.state, a.losses = gn['nodes_backup'].data_ptr(), gn['kld_backup'].data_ptr(), gn['state'].data_ptr(), gn['losses'].data_ptr()
                a.n_spans, a.n_edges, a.n_nodes, a.n_blocks = w.n_spans, w.n_edges, w.n_nodes, w.n_sources
                a.sum_N, a.max_N, a.n_unknowns, a.max_losses = gn['sum_N'], w.max_N, gn['n_y'], w.max_iters
            self._desc[level] = arr
        return self._desc[level]

    def reset_gn(self, lam=1e-4):
        for w in self.windows:
            w.reset_gn(lam)

    def run_gn(self, level, max_iters, irls_eps=1e-3, conv_tol=2e-3, pose_only=False, check_every=None, lm_up=8.0, lm_down=0.5, lm_min=1e-7,
               predicted_exit=None):
        """One Gauss-Newton phase of EVERY window (``PoseWindow.run_gn``'s arguments); ends when every window has converged or after
        ``max_iters``.  Returns the rounds launched."""
        if check_every is None:
            check_every = GN_CHECK_EVERY
        if predicted_exit is None:
            predicted_exit = GN_PREDICTED_EXIT
        for w in self.windows:
            w.begin_gn_phase()
        arr = self._records(level)
        rc = self.lib.sp_window_gn_run_multi(ctypes.addressof(arr), len(self.windows), float(irls_eps), (1 if pose_only else 0) | (2 if predicted_exit else 0),
                                             float(lm_up), float(lm_down), float(lm_min), float(conv_tol), int(max_iters), int(check_every),
                                             _lib.ptr(self.args_dev), _lib.ptr(self.states_dev), self.states_host.data_ptr(), _lib.stream_ptr())
        if rc < 0:
            _lib.check(rc, "sp_window_gn_run_multi")
        for i, w in enumerate(self.windows):
            w._gn['state_host'].copy_(self.states_host[16 * i: 16 * i + 16])
        return rc
