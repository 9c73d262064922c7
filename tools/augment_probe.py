#!/usr/bin/env python3
"""Throughput of the device-side LUNA augmentation (pcrlv2_amd/data.py -> csrc/augment.hip) at the BASELINE batch: b = 32 crops
= 64 global 64x64x32 views + 192 local 16^3 views per batch.  Prints ms per batch and crops/s, whole pipeline and per transform."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import data as D  # noqa: E402

dev = "cuda"
b = 32
pair, loc = torch.rand(b, 2, 64, 64, 32, device=dev), torch.rand(b, 6, 16, 16, 16, device=dev)
aug = D.GpuLunaAugment(dev, seed=0)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


ms = timed(lambda: aug(pair, loc))
print(f"whole batch (b={b}: 64 global + 192 local views): {ms:.2f} ms -> {b / ms * 1e3:.0f} crops/s")
v = pair.reshape(2 * b, 64, 64, 32)
g = aug.gen
flip, inv = D.draw_spatial(g, 2 * b, dev)
sigma, nstd, gamma, seed = D.draw_intensity(g, 2 * b, dev)
orig = D.draw_swap(g, 2 * b, (64, 64, 32), dev)
print(f"  global views: spatial (min + flip/affine) {timed(lambda: D.apply_spatial(v, flip, inv)):.3f} ms, "
      f"intensity without swaps {timed(lambda: D.apply_intensity(v, sigma, nstd, gamma, seed)):.3f} ms, "
      f"with 100 swaps {timed(lambda: D.apply_intensity(v, sigma, nstd, gamma, seed, orig)):.3f} ms, "
      f"parameter draws {timed(lambda: (D.draw_spatial(g, 2 * b, dev), D.draw_intensity(g, 2 * b, dev), D.draw_swap(g, 2 * b, (64, 64, 32), dev))):.3f} ms")
