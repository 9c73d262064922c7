"""2D path (BASELINE configs[4] = C5: 512x512, b=64 per GPU) step timing -- a probe, not the headline bench.

    python tools/bench_2d.py [--b 64] [--size 512] [--steps 10] [--warmup 3] [--dtype bf16]

One step = train_2d.train_step: 2 global views + 6 local 96x96 views per crop through PCRLv2 fwd+bwd, losses, fused SGD.
Prints one JSON line (crops/s, ms/step, analytic conv TFLOP/step and the MFMA fraction that implies).
"""
import argparse
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def conv_flops_fwd(size):
    """analytic forward conv FLOPs (2*MAC) of one [3,size,size] view through encoder + decoder + heads"""
    f = 0
    h = size // 2
    f += 2 * h * h * 64 * 3 * 49                                  # stem
    h //= 2
    c = 64
    for li, co in enumerate((64, 128, 256, 512)):
        s = 1 if li == 0 else 2
        ho = h // s
        f += 2 * ho * ho * co * c * 9 + 2 * ho * ho * co * co * 9  # block 0
        if s == 2:
            f += 2 * ho * ho * co * c
        f += 2 * (2 * ho * ho * co * co * 9)                       # block 1
        h, c = ho, co
    for co in (256, 128, 64, 32, 16):
        h *= 2
        f += 2 * h * h * co * c * 9 + 2 * h * h * co * co * 9      # conv1, conv2
        f += 2 * h * h * co * co * 9 + 2 * h * h * 3 * co          # deep-supervision head
        c = co
    f += 2 * h * h * 3 * 16 * 9                                    # segmentation head
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    a = ap.parse_args()
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.models import PCRLv2
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv2().cuda().set_compute_dtype(a.dtype)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator(device=dev).manual_seed(1234)
    kw = dict(generator=g, device=dev)
    x1 = torch.randn(a.b, 3, a.size, a.size, **kw)
    batch = (x1, x1 + 0.1 * torch.randn(a.b, 3, a.size, a.size, **kw), torch.rand(a.b, 3, a.size, a.size, **kw), None,
             [torch.randn(a.b, 3, 96, 96, **kw) for _ in range(6)])
    crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
    for _ in range(a.warmup):
        train_2d.train_step(model, opt, batch, 0, crit, cos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = train_2d.train_step(model, opt, batch, 0, crit, cos)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flop = 3 * a.b * (2 * conv_flops_fwd(a.size) + 6 * conv_flops_fwd(96))      # fwd + dgrad + wgrad ~ 3x forward
    print(json.dumps({"metric": "2D crops/sec pretrain step", "value": round(a.b / dt, 2), "unit": "crops/s", "ms_per_step": round(dt * 1e3, 2),
                      "dtype": a.dtype, "config": {"workload": f"PCRLv2 ResNet-18 U-Net, {a.size}x{a.size} x2 + 6 local 96x96, b={a.b}, fwd+bwd+SGD"},
                      "conv_tflop_per_step": round(flop / 1e12, 2), "step_mfma_frac": round(flop / dt / 2.5e15, 4),
                      "final_loss": float(out[0]), "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 1)}))


if __name__ == "__main__":
    main()
