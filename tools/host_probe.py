#!/usr/bin/env python3
"""How long does the HOST need to enqueue one training step (GPU idle-waiting excluded)?  + cProfile of the enqueue."""
import cProfile, pstats, io, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
torch.manual_seed(0); random.seed(0)
model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
batch = synthetic_batch(b, (64, 64, 32), 16, dev, 1)
crit, cos = MSELoss(), CosineSimilarityMean()
for _ in range(3):
    train_step(model, opt, batch, 0, crit, cos, guard=False)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step(model, opt, batch, 0, crit, cos, guard=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print("host enqueue ms/step:", [round(h, 1) for h in host], " wall ms/step:", [round(w, 1) for w in wall])
pr = cProfile.Profile()
pr.enable()
train_step(model, opt, batch, 0, crit, cos, guard=False)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:4500])
