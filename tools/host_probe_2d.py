#!/usr/bin/env python3
"""2D step (C5 per-GPU workload): how long does the HOST need to enqueue one step, and where (cProfile)?"""
import cProfile, io, os, pstats, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcrlv2_amd import train_2d
from pcrlv2_amd.models import PCRLv2
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda")
torch.manual_seed(0); random.seed(0)
model = PCRLv2().cuda().set_compute_dtype("bf16")
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
g = torch.Generator(device=dev).manual_seed(1234)
kw = dict(generator=g, device=dev)
x1 = torch.randn(b, 3, size, size, **kw)
batch = (x1, x1 + 0.1 * torch.randn(b, 3, size, size, **kw), torch.rand(b, 3, size, size, **kw), None, [torch.randn(b, 3, 96, 96, **kw) for _ in range(6)])
crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
for _ in range(3):
    train_2d.train_step(model, opt, batch, 0, crit, cos)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_2d.train_step(model, opt, batch, 0, crit, cos)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print("host enqueue ms/step:", [round(h, 1) for h in host], " wall ms/step:", [round(w, 1) for w in wall])
n0 = getattr(__import__("pcrlv2_amd._lib", fromlist=["lib"]).lib(), "ncalls", None)
pr = cProfile.Profile()
pr.enable()
train_2d.train_step(model, opt, batch, 0, crit, cos)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
