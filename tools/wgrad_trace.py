#!/usr/bin/env python3
"""Where a weight-gradient block spends its cycles (s_memtime accounting of a -DWB_TRACE=1 build of wgrad_brick.hip):

    tools/build_variant.sh trace wgrad_brick.hip -DWB_TRACE=1
    PCRL_LIB=build/var/libpcrl_trace.so python tools/wgrad_trace.py [--layers up64.1,...]

Per layer: s_memtime ticks per brick (block average), the share spent in the end-of-brick `s_waitcnt vmcnt(0)` + barrier, the share of the
request-issue window (brick start .. the step that issues the last staging request), bricks per block."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import LIBPATH, dtype_code, lib, stream_handle  # noqa: E402
from conv_probe import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=32)
    ap.add_argument("--layers", default="down64.1,up64.1,down128.1,up128.0,up128.1,up256.0,up256.1")
    args = ap.parse_args()
    L, dev, dt = lib(), torch.device("cuda"), torch.bfloat16
    raw = ctypes.CDLL(LIBPATH)
    raw.pcrl_debug_wb_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    out = (ctypes.c_ulonglong * 32)()
    sel = set(args.layers.split(","))
    for name, Ci, Co, (D, H, W) in LAYERS:
        if name not in sel:
            continue
        N = args.b
        x = ops.new_act(N, D, H, W, Ci, dt, dev).normal_()
        dy = ops.new_act(N, D, H, W, Co, dt, dev).normal_()
        dw = torch.empty(Co, Ci, 3, 3, 3, device=dev)
        nb = L.call("pcrl_conv3d_k3_wgrad_ws_bytes", N, D, H, W, Ci, Co)
        ws = ops.workspace(nb, dev)
        s = stream_handle()
        for _ in range(3):
            L.call("pcrl_conv3d_k3_wgrad", x, dy, dw, ws, nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
        assert raw.pcrl_debug_wb_trace(out) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("pcrl_conv3d_k3_wgrad", x, dy, dw, ws, nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
        e1.record()
        torch.cuda.synchronize()
        assert raw.pcrl_debug_wb_trace(out) == 0
        tot, wait, issue, bricks, blocks = (int(out[i]) for i in range(5))
        per_wave = "  per wave vmcnt-wait / barrier-wait % of block cycles: " + " ".join(f"{100.0 * int(out[8 + 2 * w]) / max(tot, 1):4.1f}/{100.0 * int(out[9 + 2 * w]) / max(tot, 1):4.1f}" for w in range(8))
        ms = e0.elapsed_time(e1)
        print(f"{name:10s} Ci={Ci:3d} Co={Co:3d} {ms:6.3f} ms (incl. second pass) blocks {blocks:4d} bricks/block {bricks / max(blocks, 1):6.1f} "
              f"ticks/block {tot / max(blocks, 1):9.0f} ticks/brick {(tot / max(bricks, 1)):7.1f}  wait+barrier {100.0 * wait / max(tot, 1):5.1f} %  "
              f"issue window {100.0 * issue / max(tot, 1):5.1f} %  other (steps after the window, prologue, epilogue) {100.0 * (tot - wait - issue) / max(tot, 1):5.1f} %", flush=True)
        print(per_wave, flush=True)


if __name__ == "__main__":
    main()
