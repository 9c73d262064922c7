import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd import train_3d as T
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss
dev = torch.device("cuda")
torch.manual_seed(0); random.seed(0)
model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
batch = synthetic_batch(32, (64, 64, 32), 16, dev, 1)
crit, cos = MSELoss(), CosineSimilarityMean()
for _ in range(4):
    T.train_step(model, opt, batch, 0, crit, cos, guard=False)
torch.cuda.synchronize()
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T.begin_step()
    losses = T.step_losses(model, batch, 0, crit, cos)
    t1 = time.perf_counter()
    opt.zero_grad()
    t2 = time.perf_counter()
    losses[0].backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print("host ms: forward+losses %.2f  zero_grad %.2f  backward %.2f  opt.step %.2f | total host %.2f  wall %.2f" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, (t5 - t0) * 1e3))
