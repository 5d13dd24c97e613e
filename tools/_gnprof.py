import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from test_gpu_sequence import make_sequence_inputs, T
from super_primitive_amd.odometery.sequence import MonoVO
n = 30
seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
vo = MonoVO(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn", translation_thresh=0.095, window_size=5, depth_of=lambda i: T(seq[i].kld_gt))
for i in range(1, n):
    vo.step(i)
    if vo.tracker is not None and vo.supp_mapper is not None and i % 5 == 0:
        tw, mw = vo.tracker.win, vo.supp_mapper.win
        print(i, "tracker: edges", tw.n_edges, "spans", tw.n_spans, "N", tw.max_N, "n_y", tw._gn['n_y'], {k: round(v, 2) for k, v in tw.gn_profile().items()})
        print(i, "mapper: edges", mw.n_edges, "spans", mw.n_spans, "N", mw.max_N, "n_y", mw._gn['n_y'], {k: round(v, 2) for k, v in mw.gn_profile().items()})
        print("   tracker stats", tw.gn_stats(), "mapper stats", mw.gn_stats())
