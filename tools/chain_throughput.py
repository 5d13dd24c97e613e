#!/usr/bin/env python
"""Config 3 as THROUGHPUT: S independent sequences through the one-call-per-frame chain (MonoVO, engine 'gn', sp_chain_step), each on its own
HIP stream and host thread.  One chain is bound by the latency of its small dependent launches (~5 % of the chip is busy), and
the foreign call releases the interpreter lock, so sequences side by side overlap; what does NOT overlap is the per-keyframe Python (window builds,
the scheduled mapping's bookkeeping).   python tools/chain_throughput.py [n_frames] [S ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from test_gpu_sequence import make_sequence_inputs, T
from super_primitive_amd.odometery.sequence import run_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ss = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
kw = dict(engine="gn", translation_thresh=0.095, window_size=5)
inputs = [make_sequence_inputs(n, rot_scale=0.3, seed=100 + k) for k in range(max(Ss))]
run_sequence(inputs[0][1][:4], inputs[0][2], T(inputs[0][0][0].T_wc), T(inputs[0][0][0].kld_gt), engine="gn")


def one(k, out):
    seq, frames, to_kf = inputs[k]
    with torch.cuda.stream(torch.cuda.Stream()):
        o = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), depth_of=lambda i: T(seq[i].kld_gt), **kw)
        torch.cuda.current_stream().synchronize()
    out[k] = o


ref = {}
for S in Ss:
    for rep in range(2):
        out = {}
        threads = [threading.Thread(target=one, args=(k, out)) for k in range(S)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in threads: t.start()
        for t in threads: t.join()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # every sequence gives what it gives alone (the first pass of S = 1 per input is the yardstick)
    same = []
    for k in range(S):
        P = out[k]["track_poses"].cpu().numpy()
        if k not in ref:
            ref[k] = P
        same.append(float(np.abs(P - ref[k]).max()))
    print(f"S = {S}: {S * (n - 1) / dt:7.0f} frames/s aggregate ({(n - 1) / dt:6.0f} per sequence; wall, frontend included), {len(out[0]['all_kf_ids'])} keyframes per sequence; "
          f"largest pose difference from the sequence's own first run {max(same):.1e}", flush=True)
