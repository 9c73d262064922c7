#!/usr/bin/env python3
"""Per-layer A/B probe of the convolution kernels (within one process, interleaved rounds, HIP-event timing).

    python tools/conv_probe.py [--b 32] [--rounds 5] [--what fwd,wgrad] [--layers up64.0,...]

Prints, per layer shape of PCRLv23d at batch b: time and TFLOP/s of
  fwd   : pcrl_conv3d_k3_fwd  with impl 0 (auto: LDS-halo brick kernel where eligible) vs impl 1 (gather kernel)
  wgrad : pcrl_conv3d_k3_wgrad
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

LAYERS = [  # name, Ci, Co, (D, H, W) for 64x64x32 inputs
    ("down64.1", 32, 64, (64, 64, 32)), ("down128.0", 64, 64, (32, 32, 16)), ("down128.1", 64, 128, (32, 32, 16)),
    ("down256.0", 128, 128, (16, 16, 8)), ("down256.1", 128, 256, (16, 16, 8)), ("down512.0", 256, 256, (8, 8, 4)),
    ("down512.1", 256, 512, (8, 8, 4)), ("up256.0", 512, 256, (16, 16, 8)), ("up256.1", 256, 256, (16, 16, 8)),
    ("up128.0", 256, 128, (32, 32, 16)), ("up128.1", 128, 128, (32, 32, 16)), ("up64.0", 128, 64, (64, 64, 32)),
    ("up64.1", 64, 64, (64, 64, 32)),
]
LOCAL = [  # the local views' gather-kernel levels (16^3 crops, batch 6 b): --b 192 --layers loc256.0,...
    ("loc256.0", 128, 128, (4, 4, 4)), ("loc256.1", 128, 256, (4, 4, 4)), ("loc512.0", 256, 256, (2, 2, 2)), ("loc512.1", 256, 512, (2, 2, 2)),
    ("locup256.1", 256, 256, (4, 4, 4)),
]


def timed_ab(fns, rounds, inner=3):
    """Interleaved A/B: every round runs each variant (its setup, then `inner` back-to-back calls between two events).
    -> [(median ms per call, min ms per call)] per variant."""
    ts = [[] for _ in fns]
    for setup, fn in fns:
        setup()
        fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, (setup, fn) in enumerate(fns):
            setup()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / inner)
    out = []
    for t in ts:
        t.sort()
        out.append((t[len(t) // 2], t[0]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--layers", default="")
    ap.add_argument("--impls", default="0,1", help="fwd / dgrad: 0 auto (brick kernel, co-located channel tiles), 1 gather, 3 brick kernel on its 2-D grid")
    ap.add_argument("--wimpls", default="0,1", help="wgrad: 0 auto (brick kernel, XCD co-located launch), 1 gather, 2 brick kernel on its 2-D grid")
    ap.add_argument("--fill", default="randn", choices=["randn", "zeros", "ones", "relu"], help="operand data (power is data dependent)")
    args = ap.parse_args()
    L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
    what = args.what.split(",")
    sel = set(args.layers.split(",")) if args.layers else None
    impls = [int(v) for v in args.impls.split(",")]
    tot = {}
    for name, Ci, Co, (D, H, W) in LAYERS + LOCAL:
        if (sel and name not in sel) or (not sel and name.startswith("loc")):
            continue
        N = args.b
        M = N * D * H * W
        flops = 2.0 * M * 27 * Ci * Co
        x = ops.new_act(N, D, H, W, Ci, dt, dev).normal_()
        dy = ops.new_act(N, D, H, W, Co, dt, dev).normal_()
        w = torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.05
        if args.fill == "zeros":
            x.zero_(); dy.zero_(); w.zero_()
        elif args.fill == "ones":
            x.fill_(1.0); dy.fill_(1.0); w.fill_(1.0)
        elif args.fill == "relu":
            x.relu_()
        wf, wd = ops.PackedWeights("conv3").get(w, dt)
        y = ops.new_act(N, D, H, W, Co, dt, dev)
        dx = ops.new_act(N, D, H, W, Ci, dt, dev)
        st = torch.empty(((M + 127) // 128) * max(Ci, Co) * 2, dtype=torch.float32, device=dev)
        s = stream_handle()
        line = f"{name:10s} Ci={Ci:3d} Co={Co:3d} {D}x{H}x{W} M={M:8d} {flops / 1e9:8.1f} GF |"
        for kind in what:
            if kind in ("fwd", "dgrad"):
                fns = []
                for impl in impls:
                    L.debug_set_conv_impl(impl)
                    if kind == "fwd":
                        nbf = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dt))
                        wsf = ops.workspace(nbf, dev) if nbf else None
                        fn = lambda wsf=wsf, nbf=nbf: L.call("pcrl_conv3d_k3_fwd_ws", x, wf, None, y, st, wsf, nbf, N, D, H, W, Ci, Co, dtype_code(dt), s)
                    else:
                        nbd = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Co, Ci, dtype_code(dt))
                        wsd = ops.workspace(nbd, dev) if nbd else None
                        fn = lambda wsd=wsd, nbd=nbd: L.call("pcrl_conv3d_k3_fwd_ws", dy, wd, None, dx, None, wsd, nbd, N, D, H, W, Co, Ci, dtype_code(dt), s)
                    fns.append((lambda impl=impl: L.debug_set_conv_impl(impl), fn))
                res = timed_ab(fns, args.rounds)
                L.debug_set_conv_impl(0)
                for impl, (med, mn) in zip(impls, res):
                    line += f" {kind}[{impl}] {med:7.3f} ms {flops / med / 1e9:6.0f} TF |"
                    tot[(kind, impl)] = tot.get((kind, impl), 0.0) + med
            elif kind == "wgrad":
                nb = L.call("pcrl_conv3d_k3_wgrad_ws_bytes", N, D, H, W, Ci, Co)
                ws = ops.workspace(nb, dev)
                dw = torch.empty_like(w)
                fn = lambda: L.call("pcrl_conv3d_k3_wgrad", x, dy, dw, ws, nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
                wimpls = [int(v) for v in args.wimpls.split(",")]
                res = timed_ab([(lambda impl=impl: L.debug_set_wgrad_impl(impl), fn) for impl in wimpls], args.rounds)
                L.debug_set_wgrad_impl(0)
                for impl, (med, mn) in zip(wimpls, res):
                    line += f" wgrad[{impl}] {med:7.3f} ms {flops / med / 1e9:6.0f} TF |"
                    tot[("wgrad", impl)] = tot.get(("wgrad", impl), 0.0) + med
        print(line, flush=True)
    print("totals (ms, one pass over the listed layers):", {f"{k[0]}[{k[1]}]": round(v, 2) for k, v in tot.items()})


if __name__ == "__main__":
    main()
