#!/usr/bin/env python3
"""Race hunt at the bench size: K training steps (b=32, 64x64x32) with every stream switch on vs the one-stream run -- parameters, momentum
buffers and BatchNorm running statistics must be BIT-identical (same kernels, only the ordering between streams differs).
   gpurun -- 'python tools/stream_stress.py [steps] [repeats]'"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from pcrlv2_amd import config
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
batch = synthetic_batch(32, (64, 64, 32), 16, dev, 7)


def run(on):
    config.WGRAD_SIDE_STREAM_3D = config.FWD_BRANCH_STREAM = config.VIEW_STREAMS = on
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    for _ in range(steps):
        out = train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean(), guard=False)
    torch.cuda.synchronize()
    rs = torch.cat([v.flatten().float() for k, v in sorted(model.state_dict().items()) if "running" in k])
    return [float(o) for o in out], opt.flat_p.clone(), opt.flat_buf.clone(), rs


ref = run(False)
print("one stream   :", ref[0])
for r in range(repeats):
    got = run(True)
    same = [torch.equal(a, b) for a, b in zip(got[1:], ref[1:])]
    print(f"three streams #{r}:", got[0], "params/momentum/running identical:", same)
    assert got[0] == ref[0] and all(same), "streams changed the result"
print("OK")
