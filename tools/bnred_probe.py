"""Data gradient + separate BatchNorm-backward reduce pass vs the fused kernel (pcrl_conv3d_k3_dgrad_bnred), per layer pair of a C2 step.
    python tools/bnred_probe.py [--b 32]
Prints us per launch: plain data gradient, reduce pass, their sum, fused kernel."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import ACT_RELU, dtype_code, lib, stream_handle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=32)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
BF, dev = torch.bfloat16, torch.device("cuda")
L, bf = lib(), dtype_code(BF)
# (name, N, D, H, W, channels of dy, channels of dx)
SH = [("down_tr64  64x64x32 64->32", a.b, 32, 64, 64, 64, 32), ("up_tr64    64x64x32 64->64", a.b, 32, 64, 64, 64, 64),
      ("down_tr128 32x32x16 128->64", a.b, 16, 32, 32, 128, 64), ("up_tr128   32x32x16 128->128", a.b, 16, 32, 32, 128, 128),
      ("down_tr256 16x16x8 256->128", a.b, 8, 16, 16, 256, 128), ("up_tr256   16x16x8 256->256", a.b, 8, 16, 16, 256, 256),
      ("local 16^3 64->32", 6 * a.b, 16, 16, 16, 64, 32), ("local 16^3 64->64", 6 * a.b, 16, 16, 16, 64, 64)]


def timeit(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


for name, N, D, H, W, Cy, Cx in SH:
    s = stream_handle()
    M = N * D * H * W
    dy = torch.randn(N, D, H, W, Cy, device=dev).to(BF)
    yb = torch.randn(N, D, H, W, Cx, device=dev).to(BF)
    w = (torch.randn(Cy, Cx, 3, 3, 3, device=dev) * 0.05)
    _, wd = ops.PackedWeights("conv3").get(w, BF)
    dx = torch.empty(N, D, H, W, Cx, device=dev, dtype=BF)
    co = [torch.rand(Cx, device=dev) + 0.5 for _ in range(4)]
    rows = L.call("pcrl_conv3d_k3_dgrad_bnred_rows", N, D, H, W, Cy, Cx, ACT_RELU, bf)
    if not rows:
        print(f"{name}: no fused kernel")
        continue
    part = torch.empty(rows * Cx * 2, device=dev)
    rows0 = L.call("pcrl_bn_bwd_partial_rows", M)
    part0 = torch.empty(rows0 * Cx * 2, device=dev)
    t_plain = timeit(lambda: L.call("pcrl_conv3d_k3_fwd_ws", dy, wd, None, dx, None, None, 0, N, D, H, W, Cy, Cx, bf, s))
    t_red = timeit(lambda: L.call("pcrl_bn_act_bwd_reduce", dx, yb, co[0], co[1], co[2], co[3], part0, M, Cx, ACT_RELU, bf, s))
    t_fused = timeit(lambda: L.call("pcrl_conv3d_k3_dgrad_bnred", dy, wd, dx, yb, co[0], co[1], co[2], co[3], part, N, D, H, W, Cy, Cx, ACT_RELU, bf, s))
    print(f"{name:32s} plain {t_plain:7.1f}  reduce {t_red:6.1f}  sum {t_plain + t_red:7.1f}  fused {t_fused:7.1f} us   ({(t_plain + t_red - t_fused):+.1f})", flush=True)
