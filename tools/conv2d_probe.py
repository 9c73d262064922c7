#!/usr/bin/env python3
"""Per-shape probe of the 2D convolution entry points (HIP events, median of rounds).
    python tools/conv2d_probe.py [--shapes gather|all] [--what fwd,dgrad,wgrad] [--rounds 7]
Shapes are (N, Hi, Wi, Ci, Co, K, stride) of the C5 step's launches (tools/shapes_2d.py)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import ops2d  # noqa: E402

GATHER = [(64, 128, 128, 64, 128, 3, 2), (64, 64, 64, 128, 256, 3, 2), (64, 32, 32, 256, 512, 3, 2), (64, 128, 128, 64, 128, 1, 2),
          (384, 12, 12, 128, 128, 3, 1), (384, 6, 6, 256, 256, 3, 1), (384, 3, 3, 512, 512, 3, 1), (384, 24, 24, 64, 128, 3, 2)]
BRICK = [(64, 128, 128, 64, 64, 3, 1), (64, 64, 64, 128, 128, 3, 1), (64, 256, 256, 32, 32, 3, 1), (384, 24, 24, 64, 64, 3, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="gather")
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--only", type=int, default=-1, help="index into the shape list")
    a = ap.parse_args()
    dev, dt = torch.device("cuda"), torch.bfloat16
    from pcrlv2_amd import config as cfg
    cfg.WGRAD_SIDE_STREAM_2D = False          # weight gradients on the timed stream
    shapes = GATHER if a.shapes == "gather" else (BRICK if a.shapes == "brick" else GATHER + BRICK)
    if a.only >= 0:
        shapes = shapes[a.only:a.only + 1]
    for (N, Hi, Wi, Ci, Co, K, st) in shapes:
        pad = (K - 1) // 2
        x = ops2d.new_act2(N, Hi, Wi, Ci, dt, dev).normal_()
        w = torch.randn(Co, Ci, K, K, device=dev) * 0.05
        pk = ops2d.PackedConv2d()
        Ho, Wo = ops2d.out_size(Hi, K, st, pad), ops2d.out_size(Wi, K, st, pad)
        dy = ops2d.new_act2(N, Ho, Wo, Co, dt, dev).normal_()
        flops = 2.0 * N * Ho * Wo * K * K * Ci * Co
        fns = {"fwd": lambda: ops2d.conv2d_forward(x, w, None, pk, st, pad, 0, dt),
               "dgrad": lambda: ops2d.conv2d_backward(x, dy, w, pk, st, pad, 0, dt, need_dx=True),
               "wgrad": lambda: ops2d.conv2d_backward(x, dy, w, pk, st, pad, 0, dt, need_dx=False)}
        line = f"({N},{Hi},{Wi},{Ci},{Co},k{K},s{st}) {flops / 1e9:7.1f} GF |"
        res = {}
        for kind in a.what.split(","):
            fn = fns[kind]
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 3)
            ts.sort()
            res[kind] = ts[len(ts) // 2]
        if "dgrad" in res and "wgrad" in res:
            res["dgrad"] -= res["wgrad"]          # conv2d_backward(need_dx=True) runs both
        for kind, ms in res.items():
            line += f" {kind} {1e3 * ms:7.1f} us {flops / ms / 1e9:5.0f} TF |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
