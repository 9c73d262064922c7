#!/usr/bin/env python3
"""Throughput of the device-side LUNA augmentation (pcrlv2_amd/data.py) on resident b=32 raw crops: crops/s it can feed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcrlv2_amd.data import GpuLunaAugment
aug = GpuLunaAugment("cuda", 0)
pair, loc = torch.rand(32, 2, 64, 64, 32, device="cuda"), torch.rand(32, 6, 16, 16, 16, device="cuda")
for _ in range(3):
    aug(pair, loc)
torch.cuda.synchronize()
t = time.time()
for _ in range(10):
    out = aug(pair, loc)
torch.cuda.synchronize()
dt = (time.time() - t) / 10
print(f"GpuLunaAugment b=32: {dt * 1e3:.1f} ms per batch = {32 / dt:.0f} crops/s")
import pcrlv2_amd.data as D
g = aug.gen
v = torch.rand(64, 64, 64, 32, device="cuda")
for name, fn in (("flip", lambda: D.random_flip(v, g)), ("affine", lambda: D.random_affine(v, g)), ("blur", lambda: D.random_blur(v, g)),
                 ("noise", lambda: D.random_noise(v, g)), ("gamma", lambda: D.random_gamma(v, g)), ("swap", lambda: D.random_swap(v, g)),
                 ("znorm", lambda: D.z_normalize(v))):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"  {name:7s} {(time.time() - t) / 5 * 1e3:6.2f} ms on 64 global crops")
