"""Host-side profile of the config-3 chain (tests/test_gpu_sequence.py's 40-frame sequence, Gauss-Newton engine, persistent supplementary
window): cProfile by cumulative and own time, plus the per-stage seconds of run_sequence.   python tools/chain_profile.py [n_frames]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from test_gpu_sequence import make_sequence_inputs, T
from super_primitive_amd.odometery.sequence import run_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seq, frames, to_kf = make_sequence_inputs(n, rot_scale=0.3)
kw = dict(engine="gn", translation_thresh=0.095, window_size=5, depth_of=lambda i: T(seq[i].kld_gt), persistent_supp=True)
run_sequence(frames[:4], to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), engine="gn")
run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
out = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
if len(sys.argv) > 2 and sys.argv[2] == "native":        # (for rocprofv3 --kernel-trace --stats: the one-call-per-frame chain only, 5 times)
    import time
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **dict(kw, native_step=True))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"native chain: {(n - 1) / dt:.0f} frames/s wall ({1e3 * dt / (n - 1):.3f} ms per frame); stages", {k: round(1e3 * v / (n - 1), 3) for k, v in o["seconds"].items()},
              "keyframes", len(o["all_kf_ids"]), "mappings", o["n_mappings"])
    sys.exit(0)
for native in (False, True):
    o = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **dict(kw, native_step=native))
    o = run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **dict(kw, native_step=native))
    print(f"native_step={native}: stages [ms per frame]:", {k: round(1e3 * v / (n - 1), 3) for k, v in o["seconds"].items()}, "chain", round((n - 1) / sum(o["seconds"].values())), "frames/s")
sec = out["seconds"]
print("stages [ms per frame]:", {k: round(1e3 * v / (n - 1), 3) for k, v in sec.items()}, "chain", round((n - 1) / sum(sec.values())), "frames/s")
pr = cProfile.Profile()
pr.enable()
run_sequence(frames, to_kf, T(seq[0].T_wc), T(seq[0].kld_gt), **kw)
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:60]))

s = io.StringIO()
st = pstats.Stats(pr, stream=s)
for pat in ("method 'cpu'", "method 'to'", "method 'clone'", "torch.tensor", "_cuda_synchronize", "method 'item'", "method 'tolist'", "torch.zeros", "stream_ptr"):
    st.print_callers(pat)
print("\n".join(l[:220] for l in s.getvalue().splitlines() if l.strip() and "Ordered by" not in l and "Function" not in l))
