#!/usr/bin/env python3
"""Time pcrl_convt3d_k2s2_fwd / _dgrad on the three UpTransition.up_conv shapes of PCRLv23d (b=32, 64x64x32 crops)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
for name, C, (D, H, W) in (("up_tr256", 512, (8, 8, 4)), ("up_tr128", 256, (16, 16, 8)), ("up_tr64", 128, (32, 32, 16))):
    N = 32
    x = ops.new_act(N, D, H, W, C, dt, dev).normal_()
    w = torch.randn(C, C, 2, 2, 2, device=dev) * 0.05
    b = torch.randn(C, device=dev)
    wf, wd = ops.PackedWeights("convt").get(w, dt)
    y = ops.new_act(N, 2 * D, 2 * H, 2 * W, C, dt, dev)
    dx = ops.new_act(N, D, H, W, C, dt, dev)
    s = stream_handle()
    out_gb = y.numel() * 2 / 1e9
    for impl in (0, 1):
        L.debug_set_conv_impl(impl)
        for kind, fn in (("fwd", lambda: L.call("pcrl_convt3d_k2s2_fwd", x, wf, b, y, N, D, H, W, C, C, dtype_code(dt), s)),
                         ("dgrad", lambda: L.call("pcrl_convt3d_k2s2_dgrad", y, wd, dx, N, D, H, W, C, C, dtype_code(dt), s))):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            print(f"{name} C={C} {D}x{H}x{W} impl={impl} {kind}: {ts[3]:.3f} ms  ({out_gb / ts[3]:.2f} TB/s of y)")
    L.debug_set_conv_impl(0)
