#!/usr/bin/env python3
"""Does the end-of-backward gradient flush of a training step take the one-launch path (pcrl_grad_sum), and if not, why?"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pcrlv2_amd import functions as F, ops
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step

dev = torch.device("cuda", 0)
torch.manual_seed(0); random.seed(0)
model = PCRLv23d().to(dev).train(); model.set_compute_dtype(torch.bfloat16)
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
batch = bench.synthetic_batch(8, (64, 64, 32), 16, dev, 1234)
orig = F._flush_fused
names = {id(p): n for n, p in model.named_parameters()}
def spy(items):
    r = orig(items)
    why = []
    if not r:
        for p, gs in items:
            for g in gs:
                if not ops.is_shared_zero(g) and (g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != p.numel() or g.data_ptr() % 16):
                    why.append((names.get(id(p)), str(g.dtype), g.is_contiguous(), tuple(g.shape), g.data_ptr() % 16))
            if p.grad is not None: why.append((names.get(id(p)), "grad set"))
            if getattr(p, "_pcrl_gslot", None) is None: why.append((names.get(id(p)), "no slot"))
    print("flush: %d parameters, fused=%s %s" % (len(items), r, why[:6]))
    return r
F._flush_fused = spy
for _ in range(2):
    train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean(), guard=False)
torch.cuda.synchronize()
