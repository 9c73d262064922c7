#!/usr/bin/env python3
"""Generate tests/golden/e_*.npz: one training step of the bfloat16-ROUNDING-AWARE comparator (oracle/pcrlv2_bf16_emulation.py: the pinned
oracle's algorithm in float64 with the engine's bf16 rounding points) on the inputs of the reference-golden cases of the same name.

    python oracle/make_emulated.py [tag ...]        (default: e_b16_32x32x16; e_luna_b8_64x64x32 needs ~50 GB and ~1 h)

The fixtures are what tests/test_model_gpu.py holds the bf16 engine to an order tighter than it can be held to the float64 golden.  Before
anything is written the comparator is checked against the pinned oracle with the rounding switched off (identity): same losses and
gradients to float64 round-off -- the rounding points are the ONLY difference."""
import os
import random
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pcrlv2_bf16_emulation as E  # noqa: E402
import pcrlv2_oracle as O  # noqa: E402
from make_golden import OUT, sample_idx  # noqa: E402

CASES = {"e_b16_32x32x16": (16, (32, 32, 16)), "e_luna_b8_64x64x32": (8, (64, 64, 32)), "e_small_b4_32x32x16": (4, (32, 32, 16))}
LOSSES = ("loss", "loss1", "loss2", "loss4", "local_loss")


def run(mod, st32, batch, epoch, seed, track_bufs=True):
    st = OrderedDict((k, (v.double().requires_grad_(True) if not O.is_buffer(k) else v.double() if v.is_floating_point() else v)) for k, v in st32.items())
    nb = {} if track_bufs else None
    with torch.backends.mkldnn.flags(enabled=False):
        r = mod.step_losses(st, batch, epoch, random.Random(seed), nb)
        names = [k for k in st if not O.is_buffer(k)]
        grads = torch.autograd.grad(r["loss"], [st[k] for k in names], allow_unused=True)
    return r, dict(zip(names, grads)), nb


def identity_check():
    """With the rounding functions replaced by the identity the comparator IS the oracle."""
    keep = E.rb, E.rv
    E.rb = E.rv = lambda t: t
    try:
        st32 = O.fill_state(torch.float32)
        batch = tuple(t.double() if torch.is_tensor(t) else [u.double() for u in t] for t in O.fill_batch(2, (16, 16, 16), dtype=torch.float32, seed=5))
        a, ga, _ = run(E, st32, batch, 3, 0)
        b, gb, _ = run(O, st32, batch, 3, 0)
        for k in LOSSES:
            assert abs(float(a[k]) - float(b[k])) < 1e-11, (k, float(a[k]), float(b[k]))
        for k in ga:
            assert (ga[k] is None) == (gb[k] is None), k
            if ga[k] is not None:
                assert (ga[k] - gb[k]).abs().max().item() <= 1e-9 * gb[k].abs().max().item() + 1e-12, k
    finally:
        E.rb, E.rv = keep
    print("comparator with rounding off == oracle (losses 1e-11, gradients 1e-9)", flush=True)


def make(tag, epoch=3, seed=0):
    b, dhw = CASES[tag]
    torch.set_num_threads(8)
    st32 = O.fill_state(torch.float32)                      # the engine's float32 master weights
    batch = tuple(t.double() if torch.is_tensor(t) else [u.double() for u in t] for t in O.fill_batch(b, dhw, dtype=torch.float32, seed=7))
    big = b * dhw[0] * dhw[1] * dhw[2] > 400000       # the 64x64x32, b = 8 case: recompute stages in backward (memory), no buffer tracking
    E.CHECKPOINT = big
    try:
        r, grads, nb = run(E, st32, batch, epoch, seed, track_bufs=not big)
    finally:
        E.CHECKPOINT = False
    fx = OrderedDict()
    fx["meta/b"], fx["meta/dhw"], fx["meta/epoch"], fx["meta/seed"], fx["meta/batch_seed"] = np.int64(b), np.array(dhw), np.int64(epoch), np.int64(seed), np.int64(7)
    for k in LOSSES:
        fx["step0/" + k] = np.float64(float(r[k]))
    fx["step0/index2"] = np.float64(r["index2"])

    def summ(t, k):
        f = t.detach().double().reshape(-1).numpy()
        return np.float64(np.sqrt((f * f).sum())), f[sample_idx(f.size, k, 3)].copy()
    fx["fwd/out/l2"], fx["fwd/out/samples"] = summ(r["mask1"], 1024)
    for i in range(3):
        fx[f"fwd/pro{i}"], fx[f"fwd/pre{i}"] = r["dec1"][i][0].detach().numpy().copy(), r["dec1"][i][1].detach().numpy().copy()
        fx[f"fwd/mid{i}/l2"], fx[f"fwd/mid{i}/samples"] = summ(r["mid1"][i], 1024)
    for name, g in grads.items():
        if g is None:
            fx[f"grad/{name}/none"] = np.int64(1)
        else:
            fx[f"grad/{name}/l2"], fx[f"grad/{name}/samples"] = summ(g, 2048)
    for name, v in (nb or {}).items():
        fx["buf1/" + name] = v.double().numpy().copy()
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **fx)
    print(f"[{tag}] losses { {k: float(r[k]) for k in LOSSES} }; wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


def make_curve(tag="e_curve_b8_32x32x16_12steps"):
    """12 SGD steps of the rounding-aware comparator on the inputs of the reference's curve fixture (curve_b8_32x32x16_12steps.npz: same
    batches, epoch, learning rate, draw seed): float32 master weights and momentum like the engine (the comparator's arithmetic runs in float64
    on them, the update of torch.optim.SGD in float32), bf16 rounding points inside the step.  What the bf16 engine's loss curve is held to at
    1e-3 where the float64 reference curve only allows 2.5e-2 (tests/test_model_gpu.py::test_bf16_loss_curve_vs_rounding_aware_comparator)."""
    ref = np.load(os.path.join(OUT, "curve_b8_32x32x16_12steps.npz"))
    b, dhw, nsteps = int(ref["b"]), tuple(int(v) for v in ref["dhw"]), int(ref["nsteps"])
    epoch, seed, base_lr = int(ref["epoch"]), int(ref["seed"]), float(ref["base_lr"])
    lr = O.lr_at(epoch, base_lr, 240)
    torch.set_num_threads(8)
    st = O.fill_state(torch.float32)
    mom, rng, curve = {}, random.Random(seed), []
    for s in range(nsteps):
        batch = tuple(t.double() if torch.is_tensor(t) else [u.double() for u in t] for t in O.fill_batch(b, dhw, dtype=torch.float32, seed=int(ref["batch_seed0"]) + s))
        st64 = OrderedDict((k, (v.double().requires_grad_(True) if not O.is_buffer(k) else v.double() if v.is_floating_point() else v)) for k, v in st.items())
        nb = {}
        with torch.backends.mkldnn.flags(enabled=False):
            r = E.step_losses(st64, batch, epoch, rng, nb)
            names = [k for k in st64 if not O.is_buffer(k)]
            grads = torch.autograd.grad(r["loss"], [st64[k] for k in names], allow_unused=True)
        g32 = {k: (None if g is None else g.float()) for k, g in zip(names, grads)}
        st, mom = O.sgd_step(st, g32, mom, lr)                  # float32 masters, float32 momentum (FusedSGD's arithmetic)
        for k, v in nb.items():
            st[k] = v.float() if v.is_floating_point() else v
        curve.append([float(r[k]) for k in LOSSES])
        print(f"[{tag}] step {s}: " + "  ".join(f"{k} {v:+.6f}" for k, v in zip(LOSSES, curve[-1])) + "   reference: " + "  ".join(f"{v:+.6f}" for v in ref["curve"][s]), flush=True)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, curve=np.array(curve), b=np.int64(b), dhw=np.array(dhw), nsteps=np.int64(nsteps), epoch=np.int64(epoch), seed=np.int64(seed),
                        base_lr=np.float64(base_lr), batch_seed0=np.int64(int(ref["batch_seed0"])))
    print(f"[{tag}] wrote {path}")


if __name__ == "__main__":
    identity_check()
    if "--curve" in sys.argv:
        make_curve()
        sys.exit(0)
    for t in (sys.argv[1:] or ["e_b16_32x32x16"]):
        make(t)
