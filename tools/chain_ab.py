"""Developer tool: interleaved A/B of the config-3 chain (one native call per frame) in ONE process, toggling an environment switch the library reads
per call:   python tools/chain_ab.py SP_WGN_NO_INLINE [n_frames] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from test_gpu_sequence import make_sequence_inputs, T
from super_primitive_amd.odometery.sequence import run_sequence

var = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
kw = dict(engine="gn", translation_thresh=0.095, window_size=5, depth_of=lambda i: T(seq[i].kld_gt), persistent_supp=True, native_step=True)
for _ in range(2):
    run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
res = {0: [], 1: []}
trk = {0: [], 1: []}
for r in range(rounds):
    for v in (0, 1):
        if v: os.environ[var] = "1"
        else: os.environ.pop(var, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[v].append((n - 1) / dt); trk[v].append(1e3 * o["seconds"]["track"] / (n - 1))
os.environ.pop(var, None)
for v in (0, 1):
    print(f"{var}={'1' if v else 'unset'}: chain {np.median(res[v]):.0f} frames/s (min {min(res[v]):.0f}, max {max(res[v]):.0f}); track stage {np.median(trk[v]):.3f} ms per frame")
