#!/usr/bin/env python3
"""Does the caching allocator's footprint plateau?  N training steps at the bench size; reserved / allocated GB and cudaMalloc count every 100."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda")
torch.manual_seed(0); random.seed(0)
model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
batch = synthetic_batch(32, (64, 64, 32), 16, dev, 1)
for i in range(n + 1):
    train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean(), guard=False)
    if i % 100 == 0:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        print(f"step {i:5d}: reserved {st['reserved_bytes.all.current'] / 2**30:6.1f} GB  allocated {st['allocated_bytes.all.current'] / 2**30:6.1f} GB  "
              f"peak allocated {st['allocated_bytes.all.peak'] / 2**30:6.1f} GB  device mallocs {st.get('num_device_alloc', 0)}  frees {st.get('num_device_free', 0)}", flush=True)
