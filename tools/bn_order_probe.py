#!/usr/bin/env python3
"""Does walking a tensor in the opposite direction of the pass before it buy Infinity-Cache hits?  Times bn_bwd_reduce + bn_bwd_apply as a
PAIR (the reduce reads da and y front to back, the apply re-reads them) under PCRL_BN_REV / PCRL_NT_MIN_MB (set in the environment)."""
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


print("PCRL_BN_REV=%s PCRL_NT_MIN_MB=%s" % (os.environ.get("PCRL_BN_REV", "-"), os.environ.get("PCRL_NT_MIN_MB", "-")))
for M, C in ((4194304, 64), (4194304, 32), (1048576, 64), (524288, 128), (524288, 64)):
    da = torch.randn(M, C, device=dev).to(dt)
    y = torch.randn(M, C, device=dev).to(dt)
    dy = torch.empty_like(y)
    f = lambda: torch.rand(C, device=dev) + 0.5
    scale, shift, mean, rstd, k1, kB, kA = f(), f() - 1, f() - 1, f(), f(), f() * 0.01, f() * 0.01
    rows = L.call("pcrl_bn_bwd_partial_rows", M)
    part = torch.empty(rows * C * 2, device=dev)
    s, d, act = stream_handle(), dtype_code(dt), 1
    gb = M * C * 2 / 1e9

    def red():
        L.call("pcrl_bn_act_bwd_reduce", da, y, scale, shift, mean, rstd, part, M, C, act, d, s)

    def app():
        L.call("pcrl_bn_act_bwd_apply", da, y, dy, scale, shift, k1, kB, kA, M, C, act, d, s)

    def pair():
        red(); app()

    tr, ta, tp = timed(red), timed(app), timed(pair)
    print(f"M={M} C={C} ({gb*1e3:.0f} MB/tensor): reduce {tr*1e3:7.1f} us | apply {ta*1e3:7.1f} us | pair {tp*1e3:7.1f} us ({5 * gb / tp:.2f} TB/s)")
