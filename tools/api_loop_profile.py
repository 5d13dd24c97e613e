import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_primitive_amd import synth
from super_primitive_amd.core import dense_optim
from super_primitive_amd.image.keyframe import KeyFrame
from super_primitive_amd.lie.se3 import SE3, LieGroupParameter
dev = torch.device("cuda:0")
p = synth.make_pair(480, 640, 64, seed=1, overlap=4, init_sigma=0.004)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src = KeyFrame(t(p.src_image), t(p.K), t(p.logdepth_perseg), t(p.keypoints), t(p.keypoint_regions))
trg = KeyFrame(t(p.trg_image), t(p.K))
kld = torch.nn.Parameter(t(p.kld_init)); T = LieGroupParameter(SE3(t(p.pose_init)[None]))
opt = torch.optim.Adam([{'params': kld, 'lr': 1e-3}, {'params': [T], 'lr': 1e-2}], lr=1e-3)
cfg = {"mode": "colour", "collect_stats": 0}
def timeit(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
pose_fixed = t(p.pose_init)
print("cost only (no grad)      %.0f us" % timeit(lambda: dense_optim.photomeric_cost(src, trg, kld.detach(), pose_fixed, cfg)))
print("pose_to_mat              %.0f us" % timeit(lambda: T.retr().matrix()[0]))
def fb():
    out = dense_optim.photomeric_cost(src, trg, kld, T.retr().matrix()[0], cfg)
    loss = torch.sum(torch.stack([torch.mean(torch.abs(out['residual']))])); loss.backward()
print("fwd+bwd (cost+pose)      %.0f us" % timeit(fb))
def step():
    opt.step(); opt.zero_grad()
fb(); print("adam step+zero_grad      %.0f us" % timeit(lambda: (fb(), step())) )
