#!/usr/bin/env python3
"""In-process A/B of a module attribute on the three-stream C2 step: blocks of K steps alternate between the two settings (same process, same box,
same draws), HIP-event time per block.

    python tools/step_ab.py pcrlv2_amd.train_3d.LAZY_SKIPS [--steps 12] [--rounds 4] [--b 32] [--dhw 64,64,32]"""
import argparse
import importlib
import os
import random
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import synthetic_batch  # noqa: E402
from pcrlv2_amd.models import PCRLv23d  # noqa: E402
from pcrlv2_amd.optim import FusedSGD  # noqa: E402
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("attr")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--b", type=int, default=32)
ap.add_argument("--dhw", default="64,64,32")
ap.add_argument("--d", type=int, default=3, choices=[2, 3], help="2: the C5 per-GPU 2D step (512 x 512, b = 64)")
a = ap.parse_args()
modname, attr = a.attr.rsplit(".", 1)
mod = importlib.import_module(modname)
dev = torch.device("cuda")
torch.manual_seed(0)
if a.d == 2:
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.models import PCRLv2
    model = PCRLv2().to(dev).set_compute_dtype("bf16")
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g2 = torch.Generator(device=dev).manual_seed(1234)
    kw = dict(generator=g2, device=dev)
    b2, sz = 64, 512
    x1 = torch.randn(b2, 3, sz, sz, **kw)
    batch = (x1, x1 + 0.1 * torch.randn(b2, 3, sz, sz, **kw), torch.rand(b2, 3, sz, sz, **kw), None, [torch.randn(b2, 3, 96, 96, **kw) for _ in range(6)])
    crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
    _ts = train_2d.train_step
    train_step = lambda m, o, bt, ep, cr, co, guard=False: _ts(m, o, bt, ep, cr, co)      # noqa: E731
else:
    model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = synthetic_batch(a.b, tuple(int(v) for v in a.dhw.split(",")), 16, dev, 1234)
    crit, cos = MSELoss(), CosineSimilarityMean()
keep = getattr(mod, attr)
random.seed(0)
for v in (True, False, True, False):
    setattr(mod, attr, v)
    for _ in range(4):
        train_step(model, opt, batch, 0, crit, cos, guard=False)
torch.cuda.synchronize()
res = {True: [], False: []}
for r in range(a.rounds):
    st = random.getstate()
    for v in (True, False):
        setattr(mod, attr, v)
        random.setstate(st)
        for _ in range(2):
            train_step(model, opt, batch, 0, crit, cos, guard=False)
        random.setstate(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.steps):
            train_step(model, opt, batch, 0, crit, cos, guard=False)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / a.steps)
setattr(mod, attr, keep)
for v in (True, False):
    print(f"{a.attr} = {v!s:5s}: " + "  ".join(f"{t:.3f}" for t in res[v]) + f"   mean {sum(res[v]) / len(res[v]):.3f} ms per step")
print(f"difference (False - True): {sum(res[False]) / len(res[False]) - sum(res[True]) / len(res[True]):+.3f} ms per step")
