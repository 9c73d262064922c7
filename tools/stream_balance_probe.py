#!/usr/bin/env python3
"""When does each stream of the three-stream step run dry?  (No profiler: events recorded at the tail of every stream after the forward and
after the backward have been enqueued; the host runs ahead, so an event's time is when the GPU got there.)

    python tools/stream_balance_probe.py [steps=20]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch
from pcrlv2_amd import ops
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd import train_3d as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0); random.seed(0)
model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
batch = synthetic_batch(32, (64, 64, 32), 16, dev, 1234)
crit, cosine = T.MSELoss(), T.CosineSimilarityMean()
for _ in range(8):
    T.train_step(model, opt, batch, 0, crit, cosine, guard=False)
torch.cuda.synchronize()


def tails():
    out = {"main": torch.cuda.current_stream(dev)}
    s = ops._side_streams.get((dev.type, dev.index))
    if s is not None:
        out["side"] = s
    for (_, _, name), vs in ops._view_streams.items():
        out[name] = vs
    ev = {}
    for k, st in out.items():
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        ev[k] = e
    return ev


rec = []
for _ in range(steps):
    T.begin_step()
    ops.throttle_host(dev)
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    losses = T.step_losses(model, batch, 0, crit, cosine)
    f = tails()
    opt.zero_grad()
    losses[0].backward()
    b = tails()
    opt.step()
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    ops.throttle_host(dev, step_done=True)
    rec.append((e0, f, b, e1))
torch.cuda.synchronize()
agg = {}
for e0, f, b, e1 in rec[3:]:
    agg.setdefault("step", []).append(e0.elapsed_time(e1))
    for k, e in f.items():
        agg.setdefault("forward tail " + k, []).append(e0.elapsed_time(e))
    for k, e in b.items():
        agg.setdefault("backward tail " + k, []).append(e0.elapsed_time(e))
for k, v in agg.items():
    print(f"{k:24s} {sum(v) / len(v):7.2f} ms after the step's start (min {min(v):.2f} max {max(v):.2f})")
