"""2D path (BASELINE configs[4] = C5: 512x512, b=64 per GPU) step timing -- a probe, not the headline bench.

    python tools/bench_2d.py [--b 64] [--size 512] [--steps 10] [--warmup 3] [--dtype bf16]

One step = train_2d.train_step: 2 global views + 6 local 96x96 views per crop through PCRLv2 fwd+bwd, losses, fused SGD.
Prints one JSON line (crops/s, ms/step, analytic conv TFLOP/step and the MFMA fraction that implies).
"""
import argparse
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def conv_flops_fwd(size):
    """analytic forward conv FLOPs (2*MAC) of one [3,size,size] view through encoder + decoder + heads"""
    f = 0
    h = size // 2
    f += 2 * h * h * 64 * 3 * 49                                  # stem
    h //= 2
    c = 64
    for li, co in enumerate((64, 128, 256, 512)):
        s = 1 if li == 0 else 2
        ho = h // s
        f += 2 * ho * ho * co * c * 9 + 2 * ho * ho * co * co * 9  # block 0
        if s == 2:
            f += 2 * ho * ho * co * c
        f += 2 * (2 * ho * ho * co * co * 9)                       # block 1
        h, c = ho, co
    for co in (256, 128, 64, 32, 16):
        h *= 2
        f += 2 * h * h * co * c * 9 + 2 * h * h * co * co * 9      # conv1, conv2
        f += 2 * h * h * co * co * 9 + 2 * h * h * 3 * co          # deep-supervision head
        c = co
    f += 2 * h * h * 3 * 16 * 9                                    # segmentation head
    return f


class Account2D:
    """Algorithmic work of the 2D step's launches, from the arguments of the C-ABI calls (pcrlv2_amd._lib counter hook: no events, no timing):
    HBM bytes = every operand tensor of a launch read or written once (weights and statistics rows ignored), MFMA flops = 2 * MACs of the
    convolution launches on the stored channel counts.  What the step cannot go below: bytes / 8 TB/s and flops / 2.5 PFLOP/s."""

    ES = {0: 4, 1: 2}      # PCRL_F32, PCRL_BF16

    def __init__(self, lib):
        self.protos = lib.protos
        self.bytes = {}
        self.flops = {}
        self.watch = {n for n in self.protos if n in self.RULES}

    def add(self, name, args):
        a = {an: v for (_, an), v in zip(self.protos[name][1], args)}
        es = self.ES.get(a.get("dtype", 1), 2)
        b, f = self.RULES[name](a, es)
        self.bytes[name] = self.bytes.get(name, 0.0) + b
        if f:
            self.flops[name] = self.flops.get(name, 0.0) + f

    @staticmethod
    def _out(h, k, s, p):
        return (h + 2 * p - k) // s + 1

    def _fwd(a, es):
        Hl, Wl = (2 * a["Hi"], 2 * a["Wi"]) if a["up"] else (a["Hi"], a["Wi"])
        Ho, Wo = Account2D._out(Hl, a["KH"], a["stride"], a["pad"]), Account2D._out(Wl, a["KW"], a["stride"], a["pad"])
        return (a["N"] * a["Hi"] * a["Wi"] * a["CiP"] * es + a["N"] * Ho * Wo * a["Co"] * (4 if a["out_f32"] else es),
                2.0 * a["N"] * Ho * Wo * a["KH"] * a["KW"] * a["CiP"] * a["Co"])

    def _wgrad(a, es):
        return (a["N"] * a["Hi"] * a["Wi"] * a["CiP"] * es + a["N"] * a["Ho"] * a["Wo"] * a["CoP"] * es,
                2.0 * a["N"] * a["Ho"] * a["Wo"] * a["KH"] * a["KW"] * a["CiP"] * a["CoP"])

    def _s2(a, es):
        taps = ((2 if a["a"] else 1) * (2 if a["b"] else 1)) if a["KH"] == 3 else 1
        return (a["N"] * a["Ho"] * a["Wo"] * a["CoP"] * es + a["N"] * a["Hi"] * a["Wi"] * a["Ci"] * es / 4,
                2.0 * a["N"] * a["Ho"] * a["Wo"] * taps * a["Ci"] * a["CoP"])

    def _nparts(a, *names):
        return sum(1 for n in names if a.get(n) is not None)

    RULES = {
        "pcrl_conv2d_fwd": _fwd,
        "pcrl_conv2d_dgrad": lambda a, es: (a["N"] * a["Ho"] * a["Wo"] * a["CoP"] * es + a["N"] * a["Hi"] * a["Wi"] * a["Ci"] * es,
                                            2.0 * a["N"] * a["Ho"] * a["Wo"] * a["KH"] * a["KW"] * a["Ci"] * a["CoP"]),
        "pcrl_conv2d_dgrad_s2": _s2,
        "pcrl_conv2d_dgrad_up": lambda a, es: (a["N"] * 4 * a["Hc"] * a["Wc"] * a["CoP"] * es + a["N"] * a["Hc"] * a["Wc"] * a["Ci"] * es,
                                               2.0 * a["N"] * 4 * a["Hc"] * a["Wc"] * 9 * a["Ci"] * a["CoP"]),
        "pcrl_conv2d_wgrad": _wgrad,
        "pcrl_stem7_fwd": lambda a, es: (a["N"] * 3 * a["H"] * a["W"] * 4 + a["N"] * a["H"] * a["W"] // 4 * 64 * 2, 2.0 * a["N"] * a["H"] * a["W"] / 4 * 147 * 64),
        "pcrl_stem7_wgrad": lambda a, es: (a["N"] * 3 * a["H"] * a["W"] * 4 + a["N"] * a["H"] * a["W"] // 4 * 64 * 2, 2.0 * a["N"] * a["H"] * a["W"] / 4 * 147 * 64),
        "pcrl_bn_act_apply": lambda a, es: (2 * a["M"] * a["C"] * es, 0),
        "pcrl_bn_act_apply_gap": lambda a, es: (2 * a["N"] * a["S"] * a["C"] * es, 0),
        "pcrl_bn_add_relu_fwd": lambda a, es: (3 * a["M"] * a["C"] * es, 0),
        "pcrl_bn_relu_maxpool2d_3s2_fwd": lambda a, es: (a["N"] * a["H"] * a["W"] * a["C"] * es + a["N"] * (a["H"] // 2) * (a["W"] // 2) * a["C"] * (es + 1), 0),
        "pcrl_bn_act_bwd_reduce": lambda a, es: (2 * a["M"] * a["C"] * es, 0),
        "pcrl_bn_act_bwd_apply": lambda a, es: (3 * a["M"] * a["C"] * es, 0),
        "pcrl_bn_act_bwd_reduce_rowadd": lambda a, es: ((1 + Account2D._nparts(a, "da")) * a["M"] * a["C"] * es, 0),
        "pcrl_bn_act_bwd_apply_rowadd": lambda a, es: ((2 + Account2D._nparts(a, "da")) * a["M"] * a["C"] * es, 0),
        "pcrl_bn_act_bwd_reduce_sum": lambda a, es: ((1 + Account2D._nparts(a, "da", "da2")) * a["M"] * a["C"] * es, 0),
        "pcrl_bn_act_bwd_apply_sum": lambda a, es: ((2 + Account2D._nparts(a, "da", "da2")) * a["M"] * a["C"] * es, 0),
        "pcrl_relu_mask_bwd": lambda a, es: (3 * a["n"] * es, 0),
        "pcrl_relu_mask_sum_bwd": lambda a, es: (4 * a["n"] * es, 0),
        "pcrl_maxpool2d_3s2_bwd_sum": lambda a, es: (a["N"] * (a["H"] // 2) * (a["W"] // 2) * a["C"] * (2 * es + 1) + a["N"] * a["H"] * a["W"] * a["C"] * es, 0),
        "pcrl_maxpool2d_3s2_bwd": lambda a, es: (a["N"] * (a["H"] // 2) * (a["W"] // 2) * a["C"] * (es + 1) + a["N"] * a["H"] * a["W"] * a["C"] * es, 0),
        "pcrl_maxpool2d_3s2_fwd": lambda a, es: (a["N"] * a["H"] * a["W"] * a["C"] * es + a["N"] * (a["H"] // 2) * (a["W"] // 2) * a["C"] * (es + 1), 0),
        "pcrl_upsample2d_nearest2_bwd": lambda a, es: (5 * a["N"] * a["H"] * a["W"] * a["C"] * es, 0),
        "pcrl_add_relu_fwd": lambda a, es: (3 * a["n"] * es, 0),
        "pcrl_mse2d_fwd": lambda a, es: (2 * a["N"] * a["HW"] * a["C"] * 4, 0),
        "pcrl_mse2d_bwd_pad": lambda a, es: (2 * a["N"] * a["HW"] * a["C"] * 4 + a["N"] * a["HW"] * a["CP"] * es, 0),
        "pcrl_conv2d_1x1_small_bwd": lambda a, es: (2 * a["M"] * a["Ci"] * es + a["M"] * a["Co"] * 4, 0),
        "pcrl_nchw_to_nhwc_pad": lambda a, es: (a["N"] * a["HW"] * (a["C"] * 4 + a["CP"] * es), 0),
        "pcrl_sgd_step": lambda a, es: (5 * a["total"] * 4 if "total" in a else 0, 0),
    }

    def summary(self, steps, seconds_per_step):
        b = sum(self.bytes.values()) / steps
        f = sum(self.flops.values()) / steps
        top = sorted(self.bytes.items(), key=lambda kv: -kv[1])[:6]
        return {"algorithmic_GB_per_step": round(b / 1e9, 2), "achieved_TBps": round(b / seconds_per_step / 1e12, 3),
                "frac_of_8TBps": round(b / seconds_per_step / 8e12, 4),
                "floor_ms_at_8TBps": round(b / 8e12 * 1e3, 2), "executed_conv_TFLOP_per_step": round(f / 1e12, 2),
                "floor_ms_at_2.5PFLOPs": round(f / 2.5e15 * 1e3, 2),
                "top_bytes_GB": {k.replace("pcrl_", ""): round(v / steps / 1e9, 2) for k, v in top}}


def brick2d_keyfn(lib):
    """EventProfiler key function: the stride-1 3x3 convolutions that run on the LDS-halo brick kernel (forward and data gradient)."""
    def key(name, args):
        protos = lib.protos[name][1]
        a = {an: v for (_, an), v in zip(protos, args)}
        if name == "pcrl_conv2d_fwd":
            kind = lib.call("pcrl_conv2d_fwd_kind", a["N"], a["Hi"], a["Wi"], a["CiP"], a["Co"], a["KH"], a["KW"], a["stride"], a["pad"], a["up"], a["out_f32"], a["dtype"])
            b, f = Account2D._fwd(a, 2)
        else:
            kind = lib.call("pcrl_conv2d_dgrad_kind", a["N"], a["Hi"], a["Wi"], a["Ci"], a["Ho"], a["Wo"], a["CoP"], a["KH"], a["KW"], a["stride"], a["pad"], a["dtype"])
            b, f = Account2D.RULES["pcrl_conv2d_dgrad"](a, 2)
        return {1: "brick_conv_kernel<2D>", 2: "conv2d_narrow_kernel", 3: "brick16_conv_kernel<2D>"}.get(kind, "conv2d_kernel(gather)"), f
    return key


def c5_report(model, opt, batch, crit, cos, train_2d, steps=3, warmup=2, roofline=True):
    """`steps` timed steps of the C5 per-GPU workload + its accounting: (seconds per step, dict for the bench line)."""
    import time as _t
    import torch as _torch
    from pcrlv2_amd import _lib, config as _cfg
    L = _lib.lib()
    for _ in range(warmup):
        train_2d.train_step(model, opt, batch, 0, crit, cos)
    _torch.cuda.synchronize()
    acct = Account2D(L)
    L.counter = acct
    t0 = _t.perf_counter()
    for _ in range(steps):
        out = train_2d.train_step(model, opt, batch, 0, crit, cos)
    _torch.cuda.synchronize()
    dt = (_t.perf_counter() - t0) / steps
    L.counter = None
    rep = {"hbm_total": acct.summary(steps, dt)}
    if roofline:
        # the dominant kernel's rate with the chip to itself: two ONE-stream steps under HIP events (as bench.py does for the 3D kernel)
        keep = (_cfg.WGRAD_SIDE_STREAM_2D, _cfg.VIEW_STREAMS_2D)
        _cfg.WGRAD_SIDE_STREAM_2D, _cfg.VIEW_STREAMS_2D = False, False
        try:
            train_2d.train_step(model, opt, batch, 0, crit, cos)
            prof = _lib.EventProfiler({"pcrl_conv2d_fwd", "pcrl_conv2d_dgrad"}, brick2d_keyfn(L))
            _torch.cuda.synchronize()
            L.profiler = prof
            for _ in range(2):
                train_2d.train_step(model, opt, batch, 0, crit, cos)
            _torch.cuda.synchronize()
            L.profiler = None
            res = prof.results()
            dom = max(res, key=lambda k: res[k][1])
            n, ms, work = res[dom]
            rep["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(work / (ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                               "frac": round(work / (ms * 1e-3) / 1e12 / 2500.0, 4), "avg_launch_ms": round(ms / n, 4), "launches": n,
                               "ms_per_step": round(ms / 2, 3), "measured": "HIP events over 2 one-stream steps (every kernel alone on the chip)",
                               "others": {k: {"launches": v[0], "ms_per_step": round(v[1] / 2, 3), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                                          for k, v in res.items() if k != dom}}
        finally:
            _cfg.WGRAD_SIDE_STREAM_2D, _cfg.VIEW_STREAMS_2D = keep
            L.profiler = None
    return dt, rep, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-roofline", action="store_true", help="skip the extra one-stream steps (rocprofv3 runs)")
    a = ap.parse_args()
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.models import PCRLv2
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv2().cuda().set_compute_dtype(a.dtype)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)          # synthetic inputs drawn on the host, resident on the device before the first step
    kw = dict(generator=g)
    x1 = torch.randn(a.b, 3, a.size, a.size, **kw)
    batch = tuple(t.to(dev) if torch.is_tensor(t) else ([u.to(dev) for u in t] if t is not None else None) for t in
                  (x1, x1 + 0.1 * torch.randn(a.b, 3, a.size, a.size, **kw), torch.rand(a.b, 3, a.size, a.size, **kw), None,
                   [torch.randn(a.b, 3, 96, 96, **kw) for _ in range(6)]))
    crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
    dt, rep, out = c5_report(model, opt, batch, crit, cos, train_2d, steps=a.steps, warmup=a.warmup, roofline=not a.no_roofline)
    flop = 3 * a.b * (2 * conv_flops_fwd(a.size) + 6 * conv_flops_fwd(96))      # fwd + dgrad + wgrad ~ 3x forward
    print(json.dumps({"metric": "2D crops/sec pretrain step", "value": round(a.b / dt, 2), "unit": "crops/s", "ms_per_step": round(dt * 1e3, 2),
                      "dtype": a.dtype, "config": {"workload": f"PCRLv2 ResNet-18 U-Net, {a.size}x{a.size} x2 + 6 local 96x96, b={a.b}, fwd+bwd+SGD"},
                      "conv_tflop_per_step": round(flop / 1e12, 2), "step_mfma_frac": round(flop / dt / 2.5e15, 4),
                      "final_loss": float(out[0]), "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 1), **rep}))


if __name__ == "__main__":
    main()
