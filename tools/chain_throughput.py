#!/usr/bin/env python
"""Config 3 as THROUGHPUT: S independent sequences through the one-call-per-frame chain (MonoVO, engine 'gn', sp_chain_step), each on its own
HIP stream and host thread.  One chain is bound by the latency of its small dependent launches (~5 % of the chip is busy), and
the foreign call releases the interpreter lock, so sequences side by side overlap; what does NOT overlap is the per-keyframe Python (window builds,
the scheduled mapping's bookkeeping).   python tools/chain_throughput.py [n_frames] [S ...]
    python tools/chain_throughput.py --processes [n_frames] [S ...]     the same with one PROCESS per sequence (no interpreter lock between them)"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from test_gpu_sequence import make_sequence_inputs, T
from super_primitive_amd.odometery.sequence import run_sequence

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    # one sequence in its own PROCESS (python tools/chain_throughput.py --child n seed start_time repeats): the chain over and over from a common start
    n, seed, start, reps = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
    seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3, seed=seed)
    res = [(T(f.image), T(f.K), T(f.logdepth_perseg), T(f.keypoints), T(f.keypoint_regions)) for f in seq]
    from super_primitive_amd.image.keyframe import KeyFrame
    to_kf = lambda i: KeyFrame(*res[i])
    kw = dict(engine="gn", translation_thresh=0.095, window_size=5, depth_of=lambda i: T(seq[i].kld_gt))
    run_sequence(frames[:4], to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn")
    run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
    torch.cuda.synchronize()
    while time.time() < start:
        time.sleep(0.001)
    t0 = time.perf_counter()
    for _ in range(reps):
        run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
    torch.cuda.synchronize()
    print(f"CHILD {reps * (n - 1)} {time.perf_counter() - t0:.6f} {time.time():.3f}", flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--processes":
    # S sequences in S PROCESSES on the one GPU (how a deployment runs independent sequences: no interpreter lock between them)
    import subprocess
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    for S in [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]:
        start = time.time() + 45.0 + 2.0 * S                     # (children import torch, render their sequence and warm up first)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(n), str(100 + k), repr(start), "20"], stdout=subprocess.PIPE, text=True)
                 for k in range(S)]
        outs = [p.communicate()[0] for p in procs]
        rows = [[float(x) for x in l.split()[1:]] for o in outs for l in o.splitlines() if l.startswith("CHILD")]
        assert len(rows) == S, outs
        frames_done, span = sum(r[0] for r in rows), max(r[2] for r in rows) - start
        print(f"S = {S} processes: {frames_done / span:7.0f} frames/s aggregate over the common span ({span:.2f} s); per process {[round(r[0] / r[1]) for r in rows]}", flush=True)
    sys.exit(0)
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
