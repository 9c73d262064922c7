#!/usr/bin/env python3
"""Per-shape table of the 2D step's convolution launches (C5 per-GPU workload): HIP events around every pcrl_conv2d_* call of two
ONE-stream steps, keyed by entry point, kernel kind (0 gather, 1 brick, 2 narrow) and geometry.   python tools/shapes_2d.py [--b 64]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench_2d import Account2D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    from pcrlv2_amd import _lib, config as cfg, train_2d
    from pcrlv2_amd.models import PCRLv2
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv2().cuda().set_compute_dtype("bf16")
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    kw = dict(generator=g)
    x1 = torch.randn(a.b, 3, a.size, a.size, **kw)
    batch = tuple(t.to(dev) if torch.is_tensor(t) else ([u.to(dev) for u in t] if t is not None else None) for t in
                  (x1, x1 + 0.1 * torch.randn(a.b, 3, a.size, a.size, **kw), torch.rand(a.b, 3, a.size, a.size, **kw), None,
                   [torch.randn(a.b, 3, 96, 96, **kw) for _ in range(6)]))
    crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
    cfg.WGRAD_SIDE_STREAM_2D, cfg.VIEW_STREAMS_2D = False, False
    L = _lib.lib()
    for _ in range(3):
        train_2d.train_step(model, opt, batch, 0, crit, cos)
    names = {"pcrl_conv2d_fwd", "pcrl_conv2d_dgrad", "pcrl_conv2d_dgrad_s2", "pcrl_conv2d_wgrad", "pcrl_conv2d_dgrad_up"}

    def key(name, args):
        p = {an: v for (_, an), v in zip(L.protos[name][1], args)}
        kind = "-"
        if name == "pcrl_conv2d_fwd":
            kind = L.call("pcrl_conv2d_fwd_kind", p["N"], p["Hi"], p["Wi"], p["CiP"], p["Co"], p["KH"], p["KW"], p["stride"], p["pad"], p["up"], p["out_f32"], p["dtype"])
            shape = (p["N"], p["Hi"], p["Wi"], p["CiP"], p["Co"], p["KH"], p["stride"], p["up"])
        elif name == "pcrl_conv2d_dgrad":
            kind = L.call("pcrl_conv2d_dgrad_kind", p["N"], p["Hi"], p["Wi"], p["Ci"], p["Ho"], p["Wo"], p["CoP"], p["KH"], p["KW"], p["stride"], p["pad"], p["dtype"])
            shape = (p["N"], p["Hi"], p["Wi"], p["Ci"], p["CoP"], p["KH"], p["stride"], 0)
        elif name == "pcrl_conv2d_dgrad_s2":
            shape = (p["N"], p["Hi"], p["Wi"], p["Ci"], p["CoP"], p["KH"], 2, 10 * p["a"] + p["b"])
        elif name == "pcrl_conv2d_dgrad_up":
            shape = (p["N"], p["Hc"], p["Wc"], p["Ci"], p["CoP"], 3, 1, 1)
        else:
            shape = (p["N"], p["Hi"], p["Wi"], p["CiP"], p["CoP"], p["KH"], p["stride"], 0)
        b, f = Account2D.RULES[name](p, 2)
        return (name.replace("pcrl_conv2d_", ""), kind, shape), (b, f)

    class Prof(_lib.EventProfiler):
        def results(self):
            torch.cuda.synchronize()
            out = {}
            for k, work, e0, e1 in self.pending:
                r = out.setdefault(k, [0, 0.0, 0.0, 0.0])
                r[0] += 1
                r[1] += e0.elapsed_time(e1)
                r[2] += work[0]
                r[3] += work[1]
            return out

    prof = Prof(names, key)
    torch.cuda.synchronize()
    L.profiler = prof
    for _ in range(2):
        train_2d.train_step(model, opt, batch, 0, crit, cos)
    torch.cuda.synchronize()
    L.profiler = None
    res = prof.results()
    tot = sum(v[1] for v in res.values()) / 2
    print(f"# convolution launches of one C5 step, one stream (b={a.b}, {a.size}^2): {tot:.2f} ms per step in {sum(v[0] for v in res.values()) // 2} launches")
    print("%-9s %4s %-38s %5s %8s %8s %7s %7s" % ("entry", "kind", "(N,Hi,Wi,Ci,Co,K,stride,up|class)", "n", "ms/step", "us/call", "TFLOP/s", "TB/s"))
    for k, v in sorted(res.items(), key=lambda kv: -kv[1][1]):
        n, ms, b, f = v
        print("%-9s %4s %-38s %5.1f %8.3f %8.1f %7.0f %7.2f" % (k[0], k[1], str(k[2]).replace(" ", ""), n / 2, ms / 2, 1e3 * ms / n, f / (ms * 1e-3) / 1e12, b / (ms * 1e-3) / 1e12))
    by = {}
    for k, v in res.items():
        r = by.setdefault((k[0], k[1]), [0, 0.0])
        r[0] += v[0] / 2
        r[1] += v[1] / 2
    print("# by entry point and kind:", {f"{k[0]}[{k[1]}]": (n, round(ms, 2)) for k, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])})


if __name__ == "__main__":
    main()
