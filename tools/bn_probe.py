#!/usr/bin/env python3
"""Time the BatchNorm/activation streaming kernels at the big activation shapes of PCRLv23d (b=32, bf16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16


def timed(fn, n=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


SHAPES = ((4194304, 64), (4194304, 32), (524288, 128), (524288, 64), (65536, 256))
if len(sys.argv) > 1:      # "M,C;M,C;..."  (2D decoder: 16777216,16;16777216,32;4194304,64)
    SHAPES = tuple(tuple(int(v) for v in t.split(",")) for t in sys.argv[1].split(";"))
for M, C in SHAPES:
    da = torch.randn(M, C, device=dev).to(dt)
    y = torch.randn(M, C, device=dev).to(dt)
    dy = torch.empty_like(y)
    a = torch.empty_like(y)
    f = lambda: torch.rand(C, device=dev) + 0.5
    scale, shift, mean, rstd, k1, kB, kA = f(), f() - 1, f() - 1, f(), f(), f() * 0.01, f() * 0.01
    rows = L.call("pcrl_bn_bwd_partial_rows", M)
    part = torch.empty(rows * C * 2, device=dev)
    s, d, act = stream_handle(), dtype_code(dt), 1
    gb = M * C * 2 / 1e9
    t = timed(lambda: L.call("pcrl_bn_act_bwd_reduce", da, y, scale, shift, mean, rstd, part, M, C, act, d, s))
    print(f"M={M} C={C}: bwd_reduce {t*1e3:7.1f} us ({2 * gb / t:.2f} TB/s)", end="")
    t = timed(lambda: L.call("pcrl_bn_act_bwd_apply", da, y, dy, scale, shift, k1, kB, kA, M, C, act, d, s))
    print(f" | bwd_apply {t*1e3:7.1f} us ({3 * gb / t:.2f} TB/s)", end="")
    t = timed(lambda: L.call("pcrl_bn_act_apply", y, a, scale, shift, M, C, act, d, s))
    print(f" | apply {t*1e3:7.1f} us ({2 * gb / t:.2f} TB/s)")
