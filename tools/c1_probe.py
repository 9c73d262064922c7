#!/usr/bin/env python3
"""First-layer (1 -> Co) forward probe: time and TB/s of pcrl_conv3d_k3_c1_fwd at the C2 shapes (global 64x64x32 x 32, local 16^3 x 192)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
for (N, D, H, W, Co) in [(32, 64, 64, 32, 32), (192, 16, 16, 16, 32), (8, 128, 128, 64, 32)]:
    x = torch.randn(N, D, H, W, device=dev)
    w = torch.randn(Co, 1, 3, 3, 3, device=dev) * 0.1
    b = torch.randn(Co, device=dev)
    y = ops.new_act(N, D, H, W, Co, dt, dev)
    rows = L.call("pcrl_conv3d_k3_c1_stats_rows", N, D, H, W, Co, dtype_code(dt))
    st = torch.empty(rows * Co * 2, dtype=torch.float32, device=dev)
    fn = lambda: L.call("pcrl_conv3d_k3_c1_fwd", x, w, b, y, st, N, D, H, W, Co, dtype_code(dt), stream_handle())
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ts.sort()
    ms = ts[len(ts) // 2]
    byt = N * D * H * W * (4 + 2 * Co)
    print(f"c1 fwd N={N} {D}x{H}x{W} Co={Co}: {1e3 * ms:7.1f} us  {byt / ms / 1e9:5.2f} TB/s  checksum {float(y.float().sum()):.4f} {float(st.sum()):.3f}")
