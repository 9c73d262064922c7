#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference, and pin the oracle against it.

Runs only in the authoring container (needs /root/reference, which never travels to the
GPU box).  What it does:
  1. imports the reference's `models/pcrlv2_model_3d.py` by file path (torch-only), and
     `train_3d.cos_loss` / `utils.adjust_learning_rate` behind stub modules for the
     reference's absent third-party imports (segmentation_models_pytorch, apex, PIL);
  2. loads the closed-form state of `pcrlv2_oracle.fill_state` into the reference model,
     runs it in float64 with oneDNN off on `fill_batch` inputs, restating
     train_3d.py:109-151 around the imported model + imported cos_loss (the loop itself has
     hard .cuda() calls and cannot run here);
  3. asserts the functional oracle reproduces the reference (outputs, losses, every
     gradient, BN buffers, 2 SGD steps) to float64 round-off;
  4. writes compact fixtures: scalars, full [b,C] features, and for big tensors the L2 norm
     plus entries sampled at hashed indices (`sample_idx`).

Usage:  python oracle/make_golden.py
"""
import importlib.util
import os
import random
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pcrlv2_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sample_idx(n: int, k: int, seed: int) -> np.ndarray:
    """k deterministic indices into a flat tensor of n elements (shared with the tests)."""
    if n <= k:
        return np.arange(n, dtype=np.int64)
    u = (O._hash_uniform(k, seed) + 1.0) * 0.5
    return np.minimum((u * n).astype(np.int64), n - 1)


def summarize(t: torch.Tensor, k: int = 64, seed: int = 3):
    f = t.detach().double().reshape(-1).numpy()
    return dict(l2=np.float64(np.sqrt((f * f).sum())), mean=np.float64(f.mean()),
                samples=f[sample_idx(f.size, k, seed)].copy())


def _stub_modules():
    smp = types.ModuleType("segmentation_models_pytorch")
    base = types.ModuleType("segmentation_models_pytorch.base")
    mods = types.ModuleType("segmentation_models_pytorch.base.modules")
    init = types.ModuleType("segmentation_models_pytorch.base.initialization")
    init.initialize_decoder = lambda m: None
    init.initialize_head = lambda m: None
    smp.base, base.modules, base.initialization = base, mods, init
    smp.Unet = object
    for n, m in (("segmentation_models_pytorch", smp), ("segmentation_models_pytorch.base", base),
                 ("segmentation_models_pytorch.base.modules", mods),
                 ("segmentation_models_pytorch.base.initialization", init)):
        sys.modules.setdefault(n, m)
    if "PIL" not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            pil = types.ModuleType("PIL")
            pil.ImageFilter = types.ModuleType("PIL.ImageFilter")
            sys.modules["PIL"] = pil
            sys.modules["PIL.ImageFilter"] = pil.ImageFilter


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_model3d", os.path.join(REF, "models", "pcrlv2_model_3d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _stub_modules()
    sys.path.insert(0, REF)
    try:
        import train_3d as ref_train  # reference module (its `models` package resolves to the stubs + real 3D file)
        import utils as ref_utils
    finally:
        sys.path.remove(REF)
    return mod, ref_train, ref_utils


def reference_step(model, ref_train, batch, epoch, criterion, cosine):
    """train_3d.py:113-138 around the imported model and imported cos_loss (CPU, no .cuda())."""
    import math
    input1, input2, gt, _gt2, local_views = batch
    bsz = input1.size(0)
    mask1, dec1, mid1 = model(input1)
    _mask2, dec2, _ = model(input2)
    loss2, index2 = ref_train.cos_loss(cosine, dec1, dec2)
    local_loss = 0.0
    local_input = torch.cat(local_views, dim=0)
    _, lout, _ = model(local_input, local=True)
    lout = [torch.stack(t) for t in lout]
    for i in range(len(local_views)):
        tmp = [t[:, bsz * i: bsz * (i + 1)] for t in lout]
        l1, _ = ref_train.cos_loss(cosine, dec1, tmp)
        l2, _ = ref_train.cos_loss(cosine, dec2, tmp)
        local_loss += l1
        local_loss += l2
    local_loss = local_loss / (2 * len(local_views))
    loss1 = criterion(mask1, gt)
    beta = 0.5 * (1. + math.cos(math.pi * epoch / 240))
    loss4 = beta * criterion(mid1[index2], gt)
    loss = loss1 + loss2 + loss4 + local_loss
    return dict(loss=loss, loss1=loss1, loss2=loss2, loss4=loss4, local_loss=local_loss, index2=index2,
                mask1=mask1, dec1=dec1, mid1=mid1)


def close(a, b, tol, what):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-30)
    assert err <= tol * max(ref, 1.0), f"oracle != reference for {what}: max|d|={err:.3e} (ref max {ref:.3e})"
    return err


def make_case(tag, b, dhw, nsteps, refmod, ref_train, ref_utils, epoch=3, base_lr=1e-3, epochs=240, seed=0):
    torch.set_num_threads(8)
    dt = torch.float64
    st0 = O.fill_state(dt)
    batches = [O.fill_batch(b, dhw, dtype=dt, seed=7 + 100 * s) for s in range(nsteps)]

    # ---------------- reference ----------------
    model = refmod.PCRLv23d().double()
    missing = model.load_state_dict(st0, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(model.state_dict().keys()) == list(st0.keys()), "state_dict order differs from oracle layout"
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=1e-4)

    class A:  # args stand-in for utils.adjust_learning_rate
        pass
    A.lr, A.epochs = base_lr, epochs
    ref_utils.adjust_learning_rate(epoch, A, opt)
    assert abs(opt.param_groups[0]["lr"] - O.lr_at(epoch, base_lr, epochs)) < 1e-18
    criterion, cosine = torch.nn.MSELoss(), torch.nn.CosineSimilarity()
    random.seed(seed)
    ref_log, ref_first, ref_first_grads = [], None, None
    with torch.backends.mkldnn.flags(enabled=False):
        for s in range(nsteps):
            r = reference_step(model, ref_train, batches[s], epoch, criterion, cosine)
            opt.zero_grad()
            r["loss"].backward()
            if s == 0:
                ref_first = r
                ref_first_grads = {k: (None if p.grad is None else p.grad.detach().clone())
                                   for k, p in model.named_parameters()}
                ref_bufs1 = {k: v.detach().clone() for k, v in model.state_dict().items() if O.is_buffer(k)}
            opt.step()
            ref_log.append({k: float(r[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")} | {"index2": r["index2"]})
    ref_final = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # ---------------- oracle, checked against the reference ----------------
    with torch.backends.mkldnn.flags(enabled=False):
        st_fin, mom, log, g0 = O.train_steps(st0, batches, epoch, base_lr, epochs, seed)
        rng = random.Random(seed)
        nb = {}
        o0 = O.step_losses(st0, batches[0], epoch, rng, nb)
    worst = 0.0
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        for s in range(nsteps):
            assert abs(log[s][k] - ref_log[s][k]) < 1e-10, (k, s, log[s][k], ref_log[s][k])
    assert [l["index2"] for l in log] == [l["index2"] for l in ref_log]
    worst = max(worst, close(o0["mask1"], ref_first["mask1"], 1e-10, "out"))
    for i in range(3):
        worst = max(worst, close(o0["dec1"][i][0], ref_first["dec1"][i][0], 1e-9, f"pro{i}"))
        worst = max(worst, close(o0["dec1"][i][1], ref_first["dec1"][i][1], 1e-9, f"pre{i}"))
        worst = max(worst, close(o0["mid1"][i], ref_first["mid1"][i], 1e-10, f"mid{i}"))
    for k, g in ref_first_grads.items():
        assert (g is None) == (g0[k] is None), f"grad None-ness differs for {k}"
        if g is not None:
            tol = 1e-9 * g.abs().max().item() + 1e-11  # zero-gradient params carry ~1e-15 noise
            assert (g - g0[k]).abs().max().item() <= tol, f"grad {k}"
    for k, v in ref_bufs1.items():
        close(nb[k].double() if nb[k].dtype != torch.int64 else nb[k].double(), v.double(), 1e-10, k)
    for k, v in ref_final.items():
        close(st_fin[k].double(), v.double(), 1e-9, "final " + k)
    print(f"[{tag}] oracle == reference (worst fwd |d| {worst:.2e}); losses {ref_log}")

    # ---------------- fixtures ----------------
    fx = OrderedDict()
    fx["meta/b"], fx["meta/dhw"], fx["meta/nsteps"] = np.int64(b), np.array(dhw), np.int64(nsteps)
    fx["meta/epoch"], fx["meta/base_lr"], fx["meta/epochs"], fx["meta/seed"] = np.int64(epoch), np.float64(base_lr), np.int64(epochs), np.int64(seed)
    fx["meta/lr"] = np.float64(opt.param_groups[0]["lr"])
    for s, l in enumerate(ref_log):
        for k, v in l.items():
            fx[f"step{s}/{k}"] = np.float64(v)
    r = ref_first
    for k, v in summarize(r["mask1"], 256).items():
        fx[f"fwd/out/{k}"] = v
    for i in range(3):
        fx[f"fwd/pro{i}"] = r["dec1"][i][0].detach().numpy().copy()
        fx[f"fwd/pre{i}"] = r["dec1"][i][1].detach().numpy().copy()
        for k, v in summarize(r["mid1"][i], 256).items():
            fx[f"fwd/mid{i}/{k}"] = v
    for name, g in ref_first_grads.items():
        if g is None:
            fx[f"grad/{name}/none"] = np.int64(1)
            continue
        for k, v in summarize(g, 64).items():
            fx[f"grad/{name}/{k}"] = v
    for name, v in ref_bufs1.items():
        fx[f"buf1/{name}"] = v.double().numpy().copy()
    for name, v in ref_final.items():
        if O.is_buffer(name):
            continue
        for k, vv in summarize(v, 64).items():
            fx[f"final/{name}/{k}"] = vv
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[{tag}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {len(fx)} arrays)")


def make_dp(tag, b_rank, dhw, nsteps, world, refmod, ref_train, ref_utils, epoch=3, base_lr=1e-3, epochs=240, seed=0, partition="sample"):
    """The reference's multi-GPU semantics -- `nn.DataParallel(model)` (train_3d.py:54) -- run on the REAL model, on CPU: replica 0 is
    the module itself (its BatchNorm buffers persist), replicas r >= 1 are `torch.func.functional_call`s of the same module with the SAME
    parameter tensors and throw-away copies of the buffers (what `replicate` builds on every forward call); every replica normalises with
    its own batch statistics; the losses of train_3d.py:119-138 are means over the gathered batch (= the mean of the replicas' losses for
    equal chunks) with ONE scale draw per cos_loss call; one backward sums the replicas' gradients into the shared parameters; one SGD step.
    partition="sample": the local views are partitioned by sample (see oracle.train_steps_data_parallel); partition="chunk": the literal
    scatter of train_3d.py:121-123 -- the [6B] view-major tensor of the GLOBAL batch cut into `world` chunks, each replica's local forward (and its
    BatchNorm statistics) over ITS chunk, outputs gathered, losses over the gathered batch.  Asserts the oracle's restatement equal, writes
    tests/golden/<tag>.npz: per-replica losses of every step, the gradient of step 0, the parameters and replica 0's buffers after `nsteps`."""
    from torch.func import functional_call
    torch.set_num_threads(8)
    dt = torch.float64
    st0 = O.fill_state(dt)
    rank_batches = [[O.fill_batch(b_rank, dhw, dtype=dt, seed=7 + 100 * s + 1000 * r) for r in range(world)] for s in range(nsteps)]
    model = refmod.PCRLv23d().double()
    model.load_state_dict(st0, strict=True)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=1e-4)

    class A:
        pass
    A.lr, A.epochs = base_lr, epochs
    ref_utils.adjust_learning_rate(epoch, A, opt)
    criterion, cosine = torch.nn.MSELoss(), torch.nn.CosineSimilarity()
    params = dict(model.named_parameters())

    def replica(r):
        if r == 0:
            return model
        def call(x, local=False):
            bufs = {k: v.clone() for k, v in model.named_buffers()}      # replicate(): copies of the buffers as they are at this forward call
            return functional_call(model, {**params, **bufs}, (x,), {"local": local})
        return call

    random.seed(seed)
    ref_log, ref_first_grads = [], None
    with torch.backends.mkldnn.flags(enabled=False):
        for s in range(nsteps):
            draws = random.getstate()
            total, entry = 0.0, []
            if partition == "chunk":
                import math
                per_rank = rank_batches[s]
                B, nl = world * b_rank, len(per_rank[0][4])
                fw = [list(replica(r)(per_rank[r][0])) for r in range(world)]               # (mask1, dec1, mid1) of every replica: view 1 ...
                dec2 = [replica(r)(per_rank[r][1])[1] for r in range(world)]                # ... then view 2 ...
                local_input = torch.cat([torch.cat([per_rank[r][4][v] for r in range(world)], dim=0) for v in range(nl)], dim=0)   # train_3d.py:121 on the global batch
                rows = nl * B // world
                louts = [replica(r)(local_input[r * rows:(r + 1) * rows], local=True)[1] for r in range(world)]       # ... then DataParallel's scatter of :123
                gathered = [[torch.cat([louts[r][k][j] for r in range(world)], dim=0) for j in range(2)] for k in range(len(louts[0]))]
                lvo = [torch.stack(t) for t in gathered]                                      # train_3d.py:125 on the gathered outputs
                for r in range(world):
                    random.setstate(draws)
                    mask1, d1, mid1 = fw[r]
                    loss2, index2 = ref_train.cos_loss(cosine, d1, dec2[r])
                    local_loss = 0.0
                    for i in range(nl):
                        tmp = [t[:, B * i + r * b_rank: B * i + (r + 1) * b_rank] for t in lvo]    # rows of replica r's samples in `t[:, bsz * i: bsz * (i + 1)]`
                        l1, _ = ref_train.cos_loss(cosine, d1, tmp)
                        l2, _ = ref_train.cos_loss(cosine, dec2[r], tmp)
                        local_loss += l1
                        local_loss += l2
                    local_loss = local_loss / (2 * nl)
                    loss1 = criterion(mask1, per_rank[r][2])
                    beta = 0.5 * (1. + math.cos(math.pi * epoch / 240))
                    loss4 = beta * criterion(mid1[index2], per_rank[r][2])
                    res = dict(loss=loss1 + loss2 + loss4 + local_loss, loss1=loss1, loss2=loss2, loss4=loss4, local_loss=local_loss, index2=index2)
                    total = total + res["loss"] / world
                    entry.append({k: float(res[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")} | {"index2": res["index2"]})
            for r in range(world if partition != "chunk" else 0):
                random.setstate(draws)               # one draw per cos_loss call, for the whole gathered batch
                res = reference_step(replica(r), ref_train, rank_batches[s][r], epoch, criterion, cosine)
                total = total + res["loss"] / world
                entry.append({k: float(res[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")} | {"index2": res["index2"]})
            opt.zero_grad()
            total.backward()
            if s == 0:
                ref_first_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
            opt.step()
            ref_log.append(entry)
    ref_final = {k: v.detach().clone() for k, v in model.state_dict().items()}

    with torch.backends.mkldnn.flags(enabled=False):
        st_fin, _mom, log, g0 = O.train_steps_data_parallel(st0, rank_batches, epoch, base_lr, epochs, seed, partition=partition)
    for s in range(nsteps):
        for r in range(world):
            for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
                assert abs(log[s][r][k] - ref_log[s][r][k]) < 1e-10, (s, r, k, log[s][r][k], ref_log[s][r][k])
            assert log[s][r]["index2"] == ref_log[s][r]["index2"]
    for k, g in ref_first_grads.items():
        assert (g is None) == (g0[k] is None), f"grad None-ness differs for {k}"
        if g is not None:
            assert (g - g0[k]).abs().max().item() <= 1e-9 * g.abs().max().item() + 1e-11, f"grad {k}"
    for k, v in ref_final.items():
        close(st_fin[k].double(), v.double(), 1e-9, "final " + k)
    # the exchange matters: a single replica on replica 0's batches ends elsewhere
    with torch.backends.mkldnn.flags(enabled=False):
        st_single, *_ = O.train_steps(st0, [rb[0] for rb in rank_batches], epoch, base_lr, epochs, seed)
    k_big = "up_tr64.ops.1.conv1.weight"
    assert (st_single[k_big] - st_fin[k_big]).abs().max().item() > 1e-6
    print(f"[{tag}] oracle (DataParallel semantics, world {world}) == reference; losses {ref_log}")

    fx = OrderedDict()
    fx["meta/b_rank"], fx["meta/dhw"], fx["meta/nsteps"], fx["meta/world"] = np.int64(b_rank), np.array(dhw), np.int64(nsteps), np.int64(world)
    fx["meta/epoch"], fx["meta/base_lr"], fx["meta/epochs"], fx["meta/seed"] = np.int64(epoch), np.float64(base_lr), np.int64(epochs), np.int64(seed)
    fx["meta/lr"] = np.float64(opt.param_groups[0]["lr"])
    fx["meta/partition"] = np.array(partition)
    for s, entry in enumerate(ref_log):
        for r, l in enumerate(entry):
            for k, v in l.items():
                fx[f"step{s}/rank{r}/{k}"] = np.float64(v)
    for name, g in ref_first_grads.items():
        if g is None:
            fx[f"grad/{name}/none"] = np.int64(1)
            continue
        for k, v in summarize(g, 64).items():
            fx[f"grad/{name}/{k}"] = v
    for name, v in ref_final.items():
        if O.is_buffer(name):
            fx[f"final_buf/{name}"] = v.double().numpy().copy()
            continue
        for k, vv in summarize(v, 64).items():
            fx[f"final/{name}/{k}"] = vv
    path = os.path.join(OUT, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[{tag}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {len(fx)} arrays)")


def make_curve(tag, b, dhw, nsteps, refmod, ref_train, ref_utils, epoch=0, base_lr=1e-3, epochs=240, seed=5):
    """Loss curve of the REAL reference (float64, oneDNN off) over `nsteps` SGD steps on correlated synthetic views:
    the fixture behind the "loss curve within 1e-3" test (SURVEY App. C scopes what can be asserted)."""
    torch.set_num_threads(8)
    dt = torch.float64
    st0 = O.fill_state(dt)
    batches = [O.fill_batch(b, dhw, dtype=dt, seed=900 + s) for s in range(nsteps)]
    model = refmod.PCRLv23d().double()
    model.load_state_dict(st0, strict=True)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=1e-4)

    class A:
        pass
    A.lr, A.epochs = base_lr, epochs
    ref_utils.adjust_learning_rate(epoch, A, opt)
    criterion, cosine = torch.nn.MSELoss(), torch.nn.CosineSimilarity()
    random.seed(seed)
    rows = []
    with torch.backends.mkldnn.flags(enabled=False):
        for s in range(nsteps):
            r = reference_step(model, ref_train, batches[s], epoch, criterion, cosine)
            opt.zero_grad()
            r["loss"].backward()
            opt.step()
            rows.append([float(r[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")] + [float(r["index2"])])
            print(f"[{tag}] step {s}: {rows[-1]}")
    # the oracle must follow the same curve (it is what runs where the reference cannot travel)
    with torch.backends.mkldnn.flags(enabled=False):
        _, _, log, _ = O.train_steps(st0, batches, epoch, base_lr, epochs, seed)
    for s in range(nsteps):
        for i, k in enumerate(("loss", "loss1", "loss2", "loss4", "local_loss")):
            assert abs(log[s][k] - rows[s][i]) < 1e-9, (s, k, log[s][k], rows[s][i])
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), curve=np.array(rows), b=np.int64(b), dhw=np.array(dhw), nsteps=np.int64(nsteps),
                        epoch=np.int64(epoch), base_lr=np.float64(base_lr), seed=np.int64(seed), batch_seed0=np.int64(900))
    print(f"[{tag}] wrote fixture; oracle == reference over {nsteps} steps")


def make_long_curve(tag, b, dhw, nsteps, refmod, ref_train, ref_utils, epoch=0, base_lr=1e-3, epochs=240, seed=5, ema=0.9):
    """Long-horizon loss curve of the REAL reference (SURVEY App. C iv): `nsteps` SGD steps of train_3d.py:109-151 on
    correlated synthetic views (fill_batch seeds 2000 + step), run TWICE from the same state and draws:
      * float64, oneDNN off  -> the curve the engines are held to (per step: total, loss1, loss2, loss4, local_loss, index2, EMA(0.9) of the total);
      * float32, stock PyTorch (oneDNN on: the reference's own CPU path) -> how far stock float32 itself drifts from float64 over the
        same steps; its per-component maxima are stored as `stock_fp32_*` and are the yardstick for the tolerances the GPU test states;
      * float32 with oneDNN off (ATen's native convolutions: the same arithmetic width in another summation order) -> `fp32_nodnn_*`: a second
        draw of that drift (the cosine terms are chaotic: one float32 run's distance from the float64 run is a sample, not a bound).
    The oracle is NOT re-run here over the whole horizon (make_curve pins it step for step over 12 steps; the restatement is the same code)."""
    torch.set_num_threads(8)
    names = ("loss", "loss1", "loss2", "loss4", "local_loss")

    def run(dt, onednn, partial=None):
        st0 = O.fill_state(dt)
        model = refmod.PCRLv23d().to(dt)
        model.load_state_dict(st0, strict=True)
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=1e-4)

        class A:
            pass
        A.lr, A.epochs = base_lr, epochs
        ref_utils.adjust_learning_rate(epoch, A, opt)
        criterion, cosine = torch.nn.MSELoss(), torch.nn.CosineSimilarity()
        random.seed(seed)
        rows = []
        # the float64 run takes hours: its state (model, momentum, draws, rows so far) is checkpointed every 25 steps and picked up again
        ck = f"/tmp/{tag}_{str(dt).split('.')[-1]}.ckpt"
        if partial is not None and os.path.exists(ck):
            st_ = torch.load(ck, weights_only=False)
            model.load_state_dict(st_["model"]); opt.load_state_dict(st_["opt"]); random.setstate(st_["random"]); rows = st_["rows"]
            print(f"[{tag}] resumed {dt} at step {len(rows)} from {ck}", flush=True)
        with torch.backends.mkldnn.flags(enabled=onednn):
            for s in range(len(rows), nsteps):
                batch = O.fill_batch(b, dhw, dtype=dt, seed=2000 + s)
                r = reference_step(model, ref_train, batch, epoch, criterion, cosine)
                opt.zero_grad()
                r["loss"].backward()
                opt.step()
                rows.append([float(r[k].detach()) for k in names] + [float(r["index2"])])
                if s % 20 == 0 or s == nsteps - 1:
                    print(f"[{tag}] {dt} step {s}: {rows[-1]}", flush=True)
                if partial is not None and (s + 1) % 25 == 0:
                    partial(np.array(rows))
                    torch.save({"model": model.state_dict(), "opt": opt.state_dict(), "random": random.getstate(), "rows": rows}, ck)       # the float64 run takes hours: a usable (shorter) fixture exists from step 25 on
        return np.array(rows)

    def ema_of(v):
        out, e = [], v[0]
        for x in v:
            e = ema * e + (1.0 - ema) * x
            out.append(e)
        return np.array(out)

    def save(c64, c32, name=None, c32n=None):
        name = name or tag
        ref = c64 if c64 is not None else c32
        e_ref = ema_of(ref[:, 0])
        extra = {}
        if c64 is not None:
            n = min(len(c64), len(c32))
            d = np.abs(c32[:n, :5] - c64[:n, :5])
            e32 = ema_of(c32[:, 0])
            extra = dict(stock_fp32_max_abs=d.max(axis=0), stock_fp32_mean_abs=d.mean(axis=0),
                         stock_fp32_ema_max_after20=np.float64(np.abs(e32[:n] - e_ref[:n])[20:].max()) if n > 21 else np.float64(0))
            for i, k in enumerate(names):
                print(f"[{tag}] stock fp32 vs fp64 over {n} steps, {k}: max {d[:, i].max():.3e} mean {d[:, i].mean():.3e}")
            print(f"[{tag}] stock fp32 EMA(total) vs fp64 after step 20: max {float(extra['stock_fp32_ema_max_after20']):.3e}")
            if c32n is not None:
                # a SECOND float32 realisation of the reference (oneDNN off: ATen's native convolutions, another summation order): the quantities behind
                # BatchNorm1d over eight rows are chaotic, and one float32 run's distance from the float64 run is one draw of that distance, not a bound
                dn = np.abs(c32n[:n, :5] - c64[:n, :5])
                en = ema_of(c32n[:, 0])
                extra.update(fp32_nodnn_curve=c32n, fp32_nodnn_max_abs=dn.max(axis=0),
                             fp32_nodnn_ema_max_after20=np.float64(np.abs(en[:n] - e_ref[:n])[20:].max()) if n > 21 else np.float64(0))
                for i, k in enumerate(names):
                    print(f"[{tag}] fp32 (oneDNN off) vs fp64 over {n} steps, {k}: max {dn[:, i].max():.3e} mean {dn[:, i].mean():.3e}")
                print(f"[{tag}] fp32 (oneDNN off) EMA(total) vs fp64 after step 20: max {float(extra['fp32_nodnn_ema_max_after20']):.3e}")
        np.savez_compressed(os.path.join(OUT if name == tag else "/tmp", f"{name}.npz"), curve=ref, ema_total=e_ref, ema=np.float64(ema), b=np.int64(b), dhw=np.array(dhw),
                            nsteps=np.int64(len(ref)), epoch=np.int64(epoch), base_lr=np.float64(base_lr), seed=np.int64(seed), batch_seed0=np.int64(2000),
                            reference_dtype="float64, oneDNN off" if c64 is not None else "float32, stock (oneDNN on)", stock_fp32_curve=c32, **extra)

    # stock float32 first (minutes) so that a fixture exists early; the float64 run (~35 s per step on 8 cores: ATen's float64 convolutions
    # take the native im2col path) then replaces `curve`; `reference_dtype` says which one a file holds
    cache = os.path.join(OUT, f"{tag}.npz")
    if "--reuse-fp32" in sys.argv and os.path.exists(cache):
        c32 = np.load(cache, allow_pickle=True)["stock_fp32_curve"]
    else:
        c32 = run(torch.float32, True)
        save(None, c32)
    c64 = None
    if "--reuse-fp64" in sys.argv and os.path.exists(cache):        # the float64 run takes ~3 h: adding a float32 realisation must not repeat it
        fx_ = np.load(cache, allow_pickle=True)
        if str(fx_["reference_dtype"]).startswith("float64") and len(fx_["curve"]) == nsteps:
            c64 = fx_["curve"]
    if c64 is None:
        c64 = run(torch.float64, False, partial=lambda rows: save(rows, c32, tag + "_partial_fp64"))     # partial float64 curves go to /tmp; the fixture is replaced when complete
        save(c64, c32)
    c32n = run(torch.float32, False, partial=lambda rows: None)       # float32 with oneDNN off: a second float32 realisation of the reference (~70 min on 8 cores: checkpointed like the float64 run)
    save(c64, c32, c32n=c32n)
    print(f"[{tag}] wrote fixture ({nsteps} steps)")


def make_loss2d(tag, b=4, nlocal=6, epoch=3, seed=11):
    """Pin of what CAN be pinned on the 2D path: the reference's `train_2d.cos_loss` (train_2d.py:111-117, imported) and its
    loss assembly (train_2d.py:139-168, restated around the imported function: the loop itself sits inside train_pcrlv2_inner behind .cuda()
    calls) on closed-form five-scale feature lists (pcrlv2_2d_oracle.fill_loss_inputs).  The 2D MODEL stays unpinned (smp / torchvision absent).
    Stores the five losses, the scale the first draw picked, and all 1 + 2 * nlocal draws in order."""
    import math
    _stub_modules()
    sys.path.insert(0, REF)
    try:
        import train_2d as ref2d
    finally:
        sys.path.remove(REF)
    import pcrlv2_2d_oracle as O2
    feats1, feats2, feats_loc, mask1, masks1, gt = O2.fill_loss_inputs(b, nlocal, dtype=torch.float64)
    cosine, criterion = torch.nn.CosineSimilarity(), torch.nn.MSELoss()
    draws = []
    real_randint = random.randint

    def spy(a_, b_):
        v = real_randint(a_, b_)
        draws.append(v)
        return v
    random.seed(seed)
    ref2d.random.randint = spy          # the module's own `random` is the global one: record the draws cos_loss takes
    try:
        # train_2d.py:139-168 (decoder_outputs1 = feats1, ...; local_views_outputs = the model's output for the concatenated local views)
        bsz = b
        loss2, index2 = ref2d.cos_loss(cosine, feats1, feats2)
        local_loss = 0.0
        local_views_outputs = [torch.stack(t) for t in feats_loc]
        for i in range(nlocal):
            tmp = [t[:, bsz * i: bsz * (i + 1)] for t in local_views_outputs]
            l1, _ = ref2d.cos_loss(cosine, feats1, tmp)
            l2, _ = ref2d.cos_loss(cosine, feats2, tmp)
            local_loss += l1
            local_loss += l2
        local_loss = local_loss / (2 * nlocal)
        loss1 = criterion(mask1, gt)
        beta = 0.5 * (1. + math.cos(math.pi * epoch / 240))
        loss4 = beta * criterion(masks1[index2], gt)
        loss = loss1 + loss2 + local_loss + loss4
    finally:
        ref2d.random.randint = real_randint
    assert len(draws) == 1 + 2 * nlocal and draws[0] == index2
    # the 2D oracle's restatement of cos_loss must agree (it is what the GPU tests of the 2D step use)
    random.seed(seed)
    l2o, k2o = O2.cos_loss(feats1, feats2)
    assert k2o == index2 and abs(float(l2o) - float(loss2)) < 1e-12
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), loss=np.float64(loss), loss1=np.float64(loss1), loss2=np.float64(loss2), loss4=np.float64(loss4),
                        local_loss=np.float64(local_loss), index2=np.int64(index2), draws=np.array(draws, dtype=np.int64), b=np.int64(b), nlocal=np.int64(nlocal),
                        epoch=np.int64(epoch), seed=np.int64(seed))
    print(f"[{tag}] loss {float(loss):+.6f} loss1 {float(loss1):.6f} loss2 {float(loss2):+.6f} loss4 {float(loss4):.6f} local {float(local_loss):+.6f} draws {draws}")


def make_mfma_pin(tag, refmod):
    """Tight pin of the bf16 MFMA kernels.  The REAL reference modules -- `LUConv(Ci, Co, 'relu', 'bn')` and
    `UpTransition(C, C, 0, 'relu', 'bn')` of models/pcrlv2_model_3d.py -- evaluated in float64 (oneDNN off) on operands that are exactly
    representable in bfloat16 (pcrlv2_oracle.mfma_pin_*_case), so that a bf16 MFMA kernel differs from these numbers by float32 accumulation
    order only.  Per convolution case: samples of conv1's output, the BatchNorm3d running statistics the module holds after ONE training-mode
    forward (= 0.1 x the batch statistics of conv1's output: the observable the kernels' (sum, sum^2) rows are held to), and from autograd of
    the module's own conv1 on a bf16-exact upstream gradient: samples of the data gradient, samples + norm of the weight gradient, the bias
    gradient.  Per composed case: the same for `ops[0].conv1(up_conv(x))` (its output, ops[0].bn1's running statistics, the data gradient)."""
    torch.set_num_threads(8)
    out = {}
    K = 4096
    with torch.backends.mkldnn.flags(enabled=False):
        for name in O.MFMA_PIN_CONV:
            N, dhw, Ci, Co = O.MFMA_PIN_CONV[name]
            c = O.mfma_pin_conv_case(name)
            m = refmod.LUConv(Ci, Co, 'relu', 'bn').double().train()
            with torch.no_grad():
                m.conv1.weight.copy_(c["w"]); m.conv1.bias.copy_(c["b"]); m.bn1.weight.copy_(c["gamma"]); m.bn1.bias.copy_(c["beta"])
            keep = {}
            h = m.conv1.register_forward_hook(lambda mod, i, o: keep.__setitem__("y", o.detach()))
            m(c["x"])
            h.remove()
            y = keep["y"]
            xr = c["x"].clone().requires_grad_(True)
            m.zero_grad()
            m.conv1(xr).backward(c["dy"])
            fy, fdx, fdw = y.reshape(-1), xr.grad.reshape(-1), m.conv1.weight.grad.reshape(-1)
            out.update({f"{name}.y": fy[sample_idx(fy.numel(), K, 3)].numpy(), f"{name}.running_mean": m.bn1.running_mean.numpy().copy(),
                        f"{name}.running_var": m.bn1.running_var.numpy().copy(), f"{name}.dx": fdx[sample_idx(fdx.numel(), K, 4)].numpy(),
                        f"{name}.dw": fdw[sample_idx(fdw.numel(), K, 5)].numpy(), f"{name}.dw_l2": np.float64(fdw.norm()), f"{name}.dw_max": np.float64(fdw.abs().max()),
                        f"{name}.db": m.conv1.bias.grad.numpy().copy(), f"{name}.y_max": np.float64(fy.abs().max()), f"{name}.dx_max": np.float64(fdx.abs().max())})
            print(f"[{tag}] {name}: |y| max {float(fy.abs().max()):.3f}  |dx| max {float(fdx.abs().max()):.3f}  |dw| l2 {float(fdw.norm()):.3f}")
        for name in O.MFMA_PIN_UP:
            N, dhw, C = O.MFMA_PIN_UP[name]
            c = O.mfma_pin_up_case(name)
            m = refmod.UpTransition(C, C, 0, 'relu', 'bn').double().train()
            with torch.no_grad():
                m.up_conv.weight.copy_(c["w_up"]); m.up_conv.bias.copy_(c["b_up"]); m.ops[0].conv1.weight.copy_(c["w0"]); m.ops[0].conv1.bias.copy_(c["b0"])
            keep = {}
            h = m.ops[0].conv1.register_forward_hook(lambda mod, i, o: keep.__setitem__("y", o.detach()))
            m.ops[0](m.up_conv(c["x"]))     # UpTransition.forward's `self.ops(self.up_conv(x))` down to ops[0] (pcrlv2_model_3d.py:61-64; the heads' BatchNorm1d needs b >= 2)
            h.remove()
            y0 = keep["y"]
            xr = c["x"].clone().requires_grad_(True)
            m.zero_grad()
            m.ops[0].conv1(m.up_conv(xr)).backward(c["dy0"])
            fy, fdx = y0.reshape(-1), xr.grad.reshape(-1)
            out.update({f"{name}.y": fy[sample_idx(fy.numel(), K, 3)].numpy(), f"{name}.running_mean": m.ops[0].bn1.running_mean.numpy().copy(),
                        f"{name}.running_var": m.ops[0].bn1.running_var.numpy().copy(), f"{name}.dx": fdx[sample_idx(fdx.numel(), K, 4)].numpy(),
                        f"{name}.y_max": np.float64(fy.abs().max()), f"{name}.dx_max": np.float64(fdx.abs().max())})
            print(f"[{tag}] {name}: |y0| max {float(fy.abs().max()):.3f}  |dx| max {float(fdx.abs().max()):.3f}")
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"[{tag}] wrote fixture")


def make_eval(tag, b, dhw, refmod, ref_train, ref_utils):
    """Eval-mode forward of the REAL reference (model.eval(): running statistics) on a state whose buffers were moved by one oracle
    training step -- what a consumer of the checkpoint runs (README.md:48-55).  The tests rebuild the state with the oracle (it is
    deterministic) and compare the HIP eval forward with these outputs."""
    dt = torch.float64
    st0 = O.fill_state(dt)
    with torch.backends.mkldnn.flags(enabled=False):
        st1, _, _, _ = O.train_steps(st0, [O.fill_batch(b, dhw, dtype=dt, seed=31)], 0, 1e-3, 240, 0)
        x = O.fill_batch(b, dhw, dtype=dt, seed=77)[0]
        model = refmod.PCRLv23d().double()
        model.load_state_dict(st1, strict=True)
        model.eval()
        with torch.no_grad():
            out, feats, masks = model(x)
            o_out, o_feats, o_masks = O.forward(st1, x, training=False)
    close(o_out, out, 1e-10, "eval out")
    for i in range(3):
        close(o_feats[i][0], feats[i][0], 1e-9, f"eval pro{i}")
        close(o_feats[i][1], feats[i][1], 1e-9, f"eval pre{i}")
        close(o_masks[i], masks[i], 1e-10, f"eval mask{i}")
    fx = OrderedDict()
    fx["meta/b"], fx["meta/dhw"], fx["meta/state_batch_seed"], fx["meta/input_seed"] = np.int64(b), np.array(dhw), np.int64(31), np.int64(77)
    for k, v in summarize(out, 512).items():
        fx[f"out/{k}"] = v
    for i in range(3):
        fx[f"pro{i}"], fx[f"pre{i}"] = feats[i][0].numpy().copy(), feats[i][1].numpy().copy()
        for k, v in summarize(masks[i], 512).items():
            fx[f"mask{i}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **fx)
    print(f"[{tag}] oracle eval == reference eval; wrote fixture")
class chunked_convs:
    """Run every F.conv3d / F.conv_transpose3d call in batch chunks.  ATen's float64 CPU convolution (slow_conv3d) materialises the im2col
    matrix of the WHOLE batch (27*Ci x voxels x 8 B per sample: 3.6 GB per sample for up_tr64.ops.0 at 64x64x32, 29 GB at 128x128x64); a
    convolution treats the samples of a batch independently, so chunking changes no arithmetic -- `make_forward` asserts bit-identity
    against the unchunked call on a small batch before it relies on it.  The module tree, forward code, BatchNorm and losses stay the
    reference's own."""

    def __init__(self, budget_bytes=12 << 30):
        self.budget = budget_bytes

    def _wrap(self, fn, taps_of):
        budget = self.budget

        def run(x, weight, *a, **kw):
            n = x.shape[0]
            vox = x[0, 0].numel()
            per = weight.shape[1] * taps_of(weight) * vox * x.element_size()   # im2col / col2im matrix of one sample
            c = max(1, min(n, budget // max(per, 1)))
            if c >= n:
                return fn(x, weight, *a, **kw)
            return torch.cat([fn(x[i:i + c], weight, *a, **kw) for i in range(0, n, c)], 0)
        return run

    def __enter__(self):
        import torch.nn.functional as Fn
        self.Fn = Fn
        self.f_conv, self.f_convt = Fn.conv3d, Fn.conv_transpose3d
        Fn.conv3d = self._wrap(self.f_conv, lambda w: w[0, 0].numel())
        Fn.conv_transpose3d = self._wrap(self.f_convt, lambda w: w[0, 0].numel())
        return self

    def __exit__(self, *exc):
        self.Fn.conv3d, self.Fn.conv_transpose3d = self.f_conv, self.f_convt


def make_forward(tag, b, dhw, refmod, ref_train, ref_utils, epoch=3, seed=0):
    """Forward-only pin at the exact BASELINE batch (C2: b = 32 at 64x64x32; C4: b = 8 at 128x128x64): float64 backward does not fit this
    container's 62 GB at those sizes, the three forwards of train_3d.py:116-138 under torch.no_grad() do.  The REAL reference model in train
    mode (batch statistics, running statistics moved by the three passes), the imported cos_loss, float64, oneDNN off: `out`, the six feature
    tensors of view 1, the three deep-supervision maps, all five losses and the BatchNorm buffers after the step's three forwards."""
    torch.set_num_threads(8)
    dt = torch.float64
    st0 = O.fill_state(dt)
    model = refmod.PCRLv23d().double()
    model.load_state_dict(st0, strict=True)
    model.train()
    criterion, cosine = torch.nn.MSELoss(), torch.nn.CosineSimilarity()
    # chunked convolutions are the same arithmetic: bit-identical to the unchunked reference on a small batch
    with torch.backends.mkldnn.flags(enabled=False), torch.no_grad():
        xs = O.fill_batch(4, (16, 16, 16), dtype=dt, seed=3)[0]
        a = model(xs)
        model.load_state_dict(st0, strict=True)
        with chunked_convs(budget_bytes=1):
            c = model(xs)
        model.load_state_dict(st0, strict=True)
    assert torch.equal(a[0], c[0]) and all(torch.equal(u, v) for fa, fc in zip(a[1], c[1]) for u, v in zip(fa, fc)), "chunking changed bits"
    del a, c, xs
    batch = O.fill_batch(b, dhw, dtype=dt, seed=7)
    random.seed(seed)
    with torch.backends.mkldnn.flags(enabled=False), torch.no_grad(), chunked_convs():
        r = reference_step(model, ref_train, batch, epoch, criterion, cosine)
    bufs = {k: v.detach().clone() for k, v in model.state_dict().items() if O.is_buffer(k)}
    # drop the activations the reference model stashes on itself before the oracle runs (memory)
    for a in ("skip_out64", "skip_out128", "skip_out256", "out512"):
        if hasattr(model, a):
            setattr(model, a, None)
    with torch.backends.mkldnn.flags(enabled=False), torch.no_grad(), chunked_convs():
        nb = {}
        o = O.step_losses(st0, batch, epoch, random.Random(seed), nb)
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        assert abs(float(o[k]) - float(r[k])) < 1e-10, (k, float(o[k]), float(r[k]))
    assert o["index2"] == r["index2"]
    worst = close(o["mask1"], r["mask1"], 1e-10, "out")
    for i in range(3):
        worst = max(worst, close(o["dec1"][i][0], r["dec1"][i][0], 1e-9, f"pro{i}"))
        worst = max(worst, close(o["dec1"][i][1], r["dec1"][i][1], 1e-9, f"pre{i}"))
        worst = max(worst, close(o["mid1"][i], r["mid1"][i], 1e-10, f"mid{i}"))
    for k, v in bufs.items():
        close(nb[k].double(), v.double(), 1e-10, k)
    fx = OrderedDict()
    fx["meta/b"], fx["meta/dhw"], fx["meta/epoch"], fx["meta/seed"], fx["meta/batch_seed"] = np.int64(b), np.array(dhw), np.int64(epoch), np.int64(seed), np.int64(7)
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        fx[f"step0/{k}"] = np.float64(float(r[k]))
    fx["step0/index2"] = np.float64(r["index2"])
    for k, v in summarize(r["mask1"], 1024).items():
        fx[f"fwd/out/{k}"] = v
    for i in range(3):
        fx[f"fwd/pro{i}"] = r["dec1"][i][0].detach().numpy().copy()
        fx[f"fwd/pre{i}"] = r["dec1"][i][1].detach().numpy().copy()
        for k, v in summarize(r["mid1"][i], 1024).items():
            fx[f"fwd/mid{i}/{k}"] = v
    for name, v in bufs.items():
        fx[f"buf1/{name}"] = v.double().numpy().copy()
    path = os.path.join(OUT, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[{tag}] oracle == reference forward-only (worst fwd |d| {worst:.2e}); losses "
          f"{ {k: float(r[k]) for k in ('loss', 'loss1', 'loss2', 'loss4', 'local_loss')} }; wrote {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB, {len(fx)} arrays)", flush=True)


def make_variant(tag, refmod, b=3, dhw=(32, 32, 16), **kw):
    """Constructor variants the reference accepts but train_3d.py:45 never instantiates (pcrlv2_model_3d.py:15-16,22-25,98): one
    train-mode forward of the REAL model built with `kw` (act / norm / in_channels / n_class), float64, oneDNN off, and the gradients
    of O.variant_loss w.r.t. every parameter; the oracle is asserted equal first."""
    torch.set_num_threads(8)
    dt = torch.float64
    okw = dict(n_class=kw.get("n_class", 1), in_channels=kw.get("in_channels", 1), act=kw.get("act", "relu"), norm=kw.get("norm", "bn"))
    st0 = O.fill_state(dt, **okw)
    model = refmod.PCRLv23d(**kw).double()
    assert list(model.state_dict().keys()) == list(st0.keys()), "state_dict order differs from the oracle layout for " + tag
    model.load_state_dict(st0, strict=True)
    model.train()
    x = O.variant_input(b, dhw, okw["in_channels"], dt)
    with torch.backends.mkldnn.flags(enabled=False):
        out, feats, masks = model(x)
        L = O.variant_loss(out, feats, masks)
        L.backward()
        ref_grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
        ref_bufs = {k: v.detach().clone() for k, v in model.state_dict().items() if O.is_buffer(k)}
        st = OrderedDict((k, (v.clone().requires_grad_(True) if not O.is_buffer(k) else v.clone())) for k, v in st0.items())
        nb = {}
        o_out, o_feats, o_masks = O.forward(st, x, new_bufs=nb, act=okw["act"], norm=okw["norm"])
        oL = O.variant_loss(o_out, o_feats, o_masks)
        pn = [k for k in st if not O.is_buffer(k)]
        og = dict(zip(pn, torch.autograd.grad(oL, [st[k] for k in pn], allow_unused=True)))
    assert abs(float(oL) - float(L)) < 1e-12, (float(oL), float(L))
    worst = close(o_out, out, 1e-10, "out")
    for i in range(3):
        worst = max(worst, close(o_feats[i][0], feats[i][0], 1e-9, f"pro{i}"), close(o_feats[i][1], feats[i][1], 1e-9, f"pre{i}"),
                    close(o_masks[i], masks[i], 1e-10, f"mask{i}"))
    for k, g in ref_grads.items():
        assert (g is None) == (og[k] is None), f"grad None-ness differs for {k}"
        if g is not None:
            assert (g - og[k]).abs().max().item() <= 1e-9 * g.abs().max().item() + 1e-11, f"grad {k}"
    for k, v in ref_bufs.items():
        close(nb[k].double(), v.double(), 1e-10, k)
    fx = OrderedDict()
    fx["meta/b"], fx["meta/dhw"] = np.int64(b), np.array(dhw)
    fx["meta/n_class"], fx["meta/in_channels"] = np.int64(okw["n_class"]), np.int64(okw["in_channels"])
    fx["meta/act"], fx["meta/norm"] = np.array(okw["act"]), np.array(okw["norm"])
    fx["loss"] = np.float64(float(L))
    for k, v in summarize(out, 512).items():
        fx[f"fwd/out/{k}"] = v
    for i in range(3):
        fx[f"fwd/pro{i}"], fx[f"fwd/pre{i}"] = feats[i][0].detach().numpy().copy(), feats[i][1].detach().numpy().copy()
        for k, v in summarize(masks[i], 256).items():
            fx[f"fwd/mask{i}/{k}"] = v
    for name, g in ref_grads.items():
        if g is None:
            fx[f"grad/{name}/none"] = np.int64(1)
            continue
        for k, v in summarize(g, 64).items():
            fx[f"grad/{name}/{k}"] = v
    for name, v in ref_bufs.items():
        fx[f"buf1/{name}"] = v.double().numpy().copy()
    path = os.path.join(OUT, f"{tag}.npz")
    np.savez_compressed(path, **fx)
    print(f"[{tag}] {kw}: oracle == reference (worst fwd |d| {worst:.2e}, loss {float(L):+.6f}); wrote {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB, {len(fx)} arrays)", flush=True)


VARIANTS = OrderedDict([
    ("v_elu", dict(act="elu")),
    ("v_prelu", dict(act="prelu")),
    ("v_in", dict(norm="in")),
    ("v_inch3", dict(in_channels=3)),
    ("v_ncls2", dict(n_class=2)),
    ("v_all", dict(act="prelu", norm="in", in_channels=2, n_class=3)),
])


def make_init(tag, refmod, seed=7):
    """Freshly constructed reference model under torch.manual_seed(seed): per-tensor sum, |sum| and leading entries.
    A drop-in model class must consume the RNG in the same order to start from the same point (models/pcrlv2_model_3d.py:85-104)."""
    torch.manual_seed(seed)
    m = refmod.PCRLv23d()
    fx = OrderedDict()
    fx["meta/seed"] = np.int64(seed)
    for k, v in m.state_dict().items():
        f = v.detach().double().reshape(-1).numpy()
        fx[f"{k}/sum"], fx[f"{k}/abs"], fx[f"{k}/head"] = np.float64(f.sum()), np.float64(np.abs(f).sum()), f[:4].copy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **fx)
    print(f"[{tag}] {len(m.state_dict())} tensors")


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the authoring container")
    refmod, ref_train, ref_utils = load_reference()
    if "--variants" in sys.argv:
        for tag, kw in VARIANTS.items():
            make_variant(tag, refmod, **kw)
        return
    if "--data-parallel" in sys.argv:
        # nn.DataParallel semantics (train_3d.py:54) on two replicas of b = 4
        if "--chunk-only" not in sys.argv:
            make_dp("dp2_b4x2_32x32x16", 4, (32, 32, 16), 2, 2, refmod, ref_train, ref_utils)
        # the same two replicas with nn.DataParallel's LITERAL scatter of the [6B] local-view tensor (train_3d.py:121-123; PCRL_DP_LOCAL_PARTITION=chunk)
        make_dp("dp2chunk_b4x2_32x32x16", 4, (32, 32, 16), 2, 2, refmod, ref_train, ref_utils, partition="chunk")
        return
    if "--mfma-pin" in sys.argv:
        make_mfma_pin("mfma_pin", refmod)
        return
    if "--loss2d" in sys.argv:
        make_loss2d("loss2d_b4_5scales")
        make_loss2d("loss2d_b2_nl3", b=2, nlocal=3, epoch=120, seed=4)
        return
    if "--long-curve" in sys.argv:
        make_long_curve("lc_b8_32x32x16_300steps", 8, (32, 32, 16), 300, refmod, ref_train, ref_utils)
        return
    if "--forward-only" in sys.argv:
        # the exact BASELINE batches (C2, C4), forward-only (float64 backward does not fit at these sizes)
        make_forward("f_c2_b32_64x64x32", 32, (64, 64, 32), refmod, ref_train, ref_utils)
        make_forward("f_c4_b8_128x128x64", 8, (128, 128, 64), refmod, ref_train, ref_utils)
        return
    make_init("init_seed7", refmod)
    if "--init-only" in sys.argv:
        return
    make_eval("eval_b2_32x32x16", 2, (32, 32, 16), refmod, ref_train, ref_utils)
    if "--eval-only" in sys.argv:
        return
    make_case("c_small_b4_32x32x16", 4, (32, 32, 16), 2, refmod, ref_train, ref_utils)
    make_case("c_luna_b2_64x64x32", 2, (64, 64, 32), 1, refmod, ref_train, ref_utils)
    # well-conditioned (BatchNorm1d over 16 rows) and BASELINE-shaped (64x64x32) steps; b = 16 at 64x64x32 needs > 63 GB in float64
    make_case("c_b16_32x32x16", 16, (32, 32, 16), 1, refmod, ref_train, ref_utils)
    make_case("c_luna_b8_64x64x32", 8, (64, 64, 32), 1, refmod, ref_train, ref_utils)
    make_curve("curve_b8_32x32x16_12steps", 8, (32, 32, 16), 12, refmod, ref_train, ref_utils)
    # LR schedule vector (utils.py:101-114) for epochs 0..240 at lr=1e-3
    class A:
        lr, epochs = 1e-3, 240

    class Opt:
        param_groups = [dict(lr=0.0)]
    lrs = []
    for e in range(0, 241):
        ref_utils.adjust_learning_rate(e, A, Opt)
        lrs.append(Opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(OUT, "lr_schedule.npz"), lr=np.array(lrs))
    # state_dict key/shape manifest from the real model
    m = refmod.PCRLv23d()
    with open(os.path.join(OUT, "state_dict_manifest.txt"), "w") as f:
        for k, v in m.state_dict().items():
            f.write(f"{k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}\n")
    print("done")


if __name__ == "__main__":
    main()
