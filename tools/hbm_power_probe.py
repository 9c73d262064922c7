#!/usr/bin/env python3
"""Board power while an HBM-bound pass loops (what a byte from HBM costs in watts, next to what the matrix kernels draw):
pcrl_bn_act_bwd_apply on a 537 MB-per-tensor activation (reads two tensors, writes one) for --secs seconds, rocm-smi sampled from a thread."""
import argparse
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--secs", type=float, default=14.0)
ap.add_argument("--M", type=int, default=4194304)
ap.add_argument("--C", type=int, default=64)
a = ap.parse_args()
L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
M, C = a.M, a.C
da = torch.randn(M, C, device=dev).to(dt)
y = torch.randn(M, C, device=dev).to(dt)
dy = torch.empty_like(y)
f = lambda: torch.rand(C, device=dev) + 0.5
scale, shift, k1, kB, kA = f(), f() - 1, f(), f() * 0.01, f() * 0.01
s, d = stream_handle(), dtype_code(dt)
samples = []


def sampler():
    time.sleep(a.secs * 0.5)
    for _ in range(6):
        out = subprocess.run("rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'package power\\|sclk\\|mclk'", shell=True, stdout=subprocess.PIPE, text=True).stdout
        samples.append(" ; ".join(ln.split(":", 1)[1].strip() if ":" in ln else ln for ln in out.strip().splitlines()))
        time.sleep(0.7)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < a.secs:
    for _ in range(200):
        L.call("pcrl_bn_act_bwd_apply", da, y, dy, scale, shift, k1, kB, kA, M, C, 1, d, s)
    n += 200
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
th.join()
ms = e0.elapsed_time(e1)
gb = 3 * M * C * 2 / 1e9
print(f"pcrl_bn_act_bwd_apply M={M} C={C}: {n} launches, {ms / n * 1e3:.1f} us each, {gb * n / ms:.2f} TB/s")
for smp in samples:
    print("  ", smp)
