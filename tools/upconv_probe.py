#!/usr/bin/env python3
"""Per-shape probe of the composed ConvTranspose3d -> Conv3d operator (csrc/upconv_fused.hip): forward, data gradient, weight-gradient
accumulation at the six shapes of a C2 step (three decoder stages x {global views b, local views 6b}), HIP-event timing.

    python tools/upconv_probe.py [--b 32] [--rounds 7] [--what fwd,dgrad,wgrad]

TFLOP/s are on the EXECUTED flops: 2 * 64 * Ci * Co per coarse voxel (8 phases x 8 taps)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

STAGES = [("up256", 512, 256, (8, 8, 4)), ("up128", 256, 128, (16, 16, 8)), ("up64", 128, 64, (32, 32, 16))]   # coarse grid of the global views


def timed(fn, rounds, inner=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    args = ap.parse_args()
    L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
    s, dc = stream_handle(), dtype_code(dt)
    tot = {}
    for local in (False, True):
        for name, Ci, Co, (D, H, W) in STAGES:
            N = args.b * 6 if local else args.b
            if local:
                D, H, W = D // 4, H // 4, (W * 2) // 4       # 16^3 local crops: coarse grids 2^3, 4^3, 8^3
            flops = 2.0 * N * D * H * W * 64 * Ci * Co
            x = ops.new_act(N, D, H, W, Ci, dt, dev).normal_()
            dy = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, dev).normal_()
            w_up = torch.randn(Ci, Ci, 2, 2, 2, device=dev) * 0.05
            b_up = torch.randn(Ci, device=dev) * 0.05
            w0 = torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.05
            b0 = torch.zeros(Co, device=dev)
            comp = ops.ComposedUpConv()
            wf, wd, tab = comp.get(w_up, b_up, w0, b0, dt, geom=(N, D, H, W))
            y = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, dev)
            dx = ops.new_act(N, D, H, W, Ci, dt, dev)
            rows = L.call("pcrl_upconv_stats_rows", N, D, H, W, Ci, Co, dc)
            st = torch.empty(rows * Co * 2, dtype=torch.float32, device=dev)
            dweff, box = torch.zeros(64 * Ci * Co, device=dev), torch.zeros(27 * Co, device=dev)
            nb = L.call("pcrl_upconv_wgrad_accum_ws_bytes", N, D, H, W, Ci, Co, dc)
            ws = ops.workspace(nb, dev)
            nbd = L.call("pcrl_upconv_dgrad_ws_bytes", N, D, H, W, Ci, Co, dc)
            wsd = torch.empty(max(nbd, 16), dtype=torch.uint8, device=dev) if nbd else None
            fb = bool(L.call("pcrl_upconv_fwd_uses_brick", N, D, H, W, Ci, Co, dc))
            db = bool(L.call("pcrl_upconv_dgrad_uses_brick", N, D, H, W, Ci, Co, dc))
            fns = {"fwd": lambda: L.call("pcrl_upconv_fwd", x, wf, comp.w3f, tab, y, st, N, D, H, W, Ci, Co, dc, s),
                   "dgrad": lambda: L.call("pcrl_upconv_dgrad_ws", dy, wd, comp.wd3, dx, wsd, nbd, N, D, H, W, Ci, Co, dc, s),
                   "wgrad": lambda: L.call("pcrl_upconv_wgrad_accum", x, dy, dweff, box, 3, ws, nb, N, D, H, W, Ci, Co, dc, s)}
            line = f"{name:6s}{' local' if local else ' global'} N={N:3d} {D}x{H}x{W} Ci={Ci} Co={Co} {flops / 1e9:7.1f} GF (brick fwd {int(fb)} dgrad {int(db)}) |"
            for k in args.what.split(","):
                ms = timed(fns[k], args.rounds)
                line += f" {k} {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF |"
                tot[k] = tot.get(k, 0.0) + ms * (1 if local else 2)
            print(line, flush=True)
    print("per step (global x2 + local x1), ms:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
