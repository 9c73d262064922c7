#!/usr/bin/env python3
"""Which host lines still launch ATen / runtime kernels inside a training step (everything else goes through libpcrl_hip.so).

    python tools/aten_probe.py [--b 32]

Runs a few warm steps, then one step under torch.profiler (CPU activity with python stacks) and prints every aten:: operator that
launched a device kernel or copy, with its count and the innermost pcrlv2_amd frame."""
import argparse
import collections
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=None)
    ap.add_argument("--d", type=int, default=3, choices=[2, 3], help="3: the 3D step (C2 shape); 2: the 2D step (C5 per-GPU shape)")
    args = ap.parse_args()
    if args.d == 2:
        return main_2d(args.b or 64)
    args.b = args.b or 32
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from pcrlv2_amd.models import PCRLv23d
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv23d().to(dev).train()
    model.set_compute_dtype(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = bench.synthetic_batch(args.b, (64, 64, 32), 16, dev, 1234)
    crit, cosine = MSELoss(), CosineSimilarityMean()
    for _ in range(4):
        train_step(model, opt, batch, 0, crit, cosine, guard=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        train_step(model, opt, batch, 0, crit, cosine, guard=False)
        torch.cuda.synchronize()
    report(prof)


def main_2d(b):
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.models import PCRLv2
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv2().cuda().set_compute_dtype("bf16")
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    kw = dict(generator=g)
    x1 = torch.randn(b, 3, 512, 512, **kw)
    batch = tuple(t.to(dev) if torch.is_tensor(t) else ([u.to(dev) for u in t] if t is not None else None) for t in
                  (x1, x1 + 0.1 * torch.randn(b, 3, 512, 512, **kw), torch.rand(b, 3, 512, 512, **kw), None,
                   [torch.randn(b, 3, 96, 96, **kw) for _ in range(6)]))
    crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
    for _ in range(4):
        train_2d.train_step(model, opt, batch, 0, crit, cos)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        train_2d.train_step(model, opt, batch, 0, crit, cos)
        torch.cuda.synchronize()
    report(prof)


def report(prof):
    agg, nk = collections.Counter(), collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        site = "?"
        for fr in ev.stack:
            if "pcrlv2_amd" in fr or "bench.py" in fr:
                site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
                site = site.split("pcrlv2_amd/")[-1] if "pcrlv2_amd/" in site else site
                break
        if site == "?":     # launched from the autograd engine's thread: the operand shapes identify the node
            site = "autograd thread, shapes " + str(ev.input_shapes)[:90]
            if os.environ.get("ATEN_PROBE_PARENTS"):     # the chain of enclosing profiler events (autograd node names)
                par, chain = ev.cpu_parent, []
                while par is not None and len(chain) < 4:
                    chain.append(par.name[:60])
                    par = par.cpu_parent
                site += " <- " + " <- ".join(chain)
        agg[(ev.name, site, ev.kernels[0].name[:50])] += 1
        nk[(ev.name, site, ev.kernels[0].name[:50])] += len(ev.kernels)
    tot = totk = 0
    for (name, site, k), n in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d} ops {nk[(name, site, k)]:4d} device launches  {name:28s} {k:52s} {site}")
        tot += n
        totk += nk[(name, site, k)]
    print("total ATen ops that launched device work in one step:", tot, "-- device launches (kernels + copies) behind them:", totk)
    # everything on the device that is NOT one of ours: runtime copies / fills / ATen kernels, by name
    dev = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and not ev.name.startswith(("void (anonymous namespace)", "(anonymous namespace)")) and "_kernel" not in ev.name:
            dev[ev.name[:70]] += 1
    for k, n in dev.most_common(12):
        print(f"   device-side event {n:4d} x {k}")


if __name__ == "__main__":
    main()
