"""GPU parity of the whole pre-training step (PCRLv23d + losses + SGD on libpcrl_hip.so) against
  (1) the golden vectors produced by the REAL reference in float64 (tests/golden/*.npz), and
  (2) the CPU oracle run live on other inputs,
plus size-independent properties at BASELINE.json's full sizes (b=32, 64x64x32, bf16).

Stated tolerances.  Calibration: stock PyTorch-CPU float32 (oneDNN off) on the same fixture differs from the float64
golden by 2e-7 (total loss, step 0), 1.2e-4 (total loss, step 1 -- the cosine terms go through BatchNorm1d over b=4
rows and amplify rounding; the MSE terms stay < 3e-6), and up to 7.0e-3 per-tensor gradient rel-L2
(`up_tr128.ops.1.bn1.bias`); see SURVEY App. C.  The HIP float32 path is held to the same envelope:
  float32 mode : sigmoid maps 5e-5 abs, [b,C] features 2e-4 abs, step-0 losses 1e-5 abs, step-1 MSE losses 2e-5 /
                 cosine losses 5e-4 abs, gradients per-tensor rel-L2 1e-2 (analytically-zero gradients: 1e-5 abs),
                 parameters after 2 SGD steps 5e-5 abs (stock torch fp32: 2.6e-5).
  bfloat16 mode: activations carry 8 mantissa bits through 17 conv+BN layers: losses 3e-2 abs, maps 3e-2 abs,
                 features: cosine similarity to the golden > 0.98.  Gradients of the FULL loss on this b=4 fixture are
                 ill-conditioned under ANY bf16 rounding: the float64 oracle with bf16 rounding emulated at the same
                 activation points is already 0.82 rel-L2 (median 0.44) away from the golden, so only gradient NORMS
                 (within 30 %) are asserted there; on the restoration (MSE) path against the live float64 oracle the
                 bf16 gradients must agree in direction (cosine > 0.7) and norm (25 %) for every weight tensor.
                 Every bf16 KERNEL is held to tight per-operator bounds in tests/test_ops_gpu.py.
"""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pcrlv2_oracle as O  # noqa: E402
from make_golden import sample_idx  # noqa: E402
from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402
from pcrlv2_amd.models import PCRLv23d  # noqa: E402
from pcrlv2_amd.optim import FusedSGD  # noqa: E402
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, cos_loss, train_step  # noqa: E402

DEV = "cuda"
ZERO_GRAD = ("conv1.bias", ".bn.bias", "predictor_head.0.bias")  # followed by a batch-stat normalisation -> exact 0


def samples(t, k, seed=3):
    f = t.detach().double().cpu().reshape(-1).numpy()
    return f[sample_idx(f.size, k, seed)]


def build(dtype, state=None):
    model = PCRLv23d().to(DEV)
    model.load_state_dict(state if state is not None else O.fill_state(torch.float32))
    model.train()
    model.set_compute_dtype(dtype)
    return model


def forward_losses(model, batch, epoch, seed):
    """train_3d.py:113-138 on the HIP path, keeping the intermediate tensors for inspection."""
    input1, input2, gt, _, local_views = batch
    crit, cosine = MSELoss(), CosineSimilarityMean()
    bsz = input1.size(0)
    random.seed(seed)
    x1, x2, gtd = input1.float().to(DEV), input2.float().to(DEV), gt.float().to(DEV)
    mask1, dec1, mid1 = model(x1)
    _, dec2, _ = model(x2)
    loss2, index2 = cos_loss(cosine, dec1, dec2)
    local_in = torch.cat([v.float().to(DEV) for v in local_views], 0)
    _, lout, _ = model(local_in, local=True)
    lout = [torch.stack(t) for t in lout]
    local_loss = 0.0
    for i in range(len(local_views)):
        tmp = [t[:, bsz * i: bsz * (i + 1)] for t in lout]
        l1, _ = cos_loss(cosine, dec1, tmp)
        l2, _ = cos_loss(cosine, dec2, tmp)
        local_loss = local_loss + l1 + l2
    local_loss = local_loss / (2 * len(local_views))
    loss1 = crit(mask1, gtd)
    import math
    beta = 0.5 * (1. + math.cos(math.pi * epoch / 240))
    loss4 = beta * crit(mid1[index2], gtd)
    loss = loss1 + loss2 + loss4 + local_loss
    return dict(loss=loss, loss1=loss1, loss2=loss2, loss4=loss4, local_loss=local_loss, index2=index2,
                mask1=mask1, dec1=dec1, mid1=mid1)


@pytest.fixture(scope="module")
def golden(golden_dir):
    fx = np.load(os.path.join(golden_dir, "c_small_b4_32x32x16.npz"))
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    batches = [O.fill_batch(b, dhw, dtype=torch.float32, seed=7 + 100 * s) for s in range(int(fx["meta/nsteps"]))]
    return fx, batches


# float32 gradient gates (5e-3 per-tensor rel-L2 against the float64 golden).  Measured on MI355X (tools/gates_probe.py, round 4):
#   c_small_b4: 70 of 71 tensors <= 4.2e-3, up_tr128.ops.1.bn1.bias 5.2e-3;   c_b16: 55 of 71 <= 5e-3, the other 16 between 5.3e-3 and 8.2e-3 --
#   every one of them an ENCODER tensor or a per-channel BatchNorm vector (down_tr64.ops.1.conv1.weight 8.2e-3, down_tr128.ops.1.bn1.bias 8.1e-3, ...).
# Why those: a BatchNorm vector entry is a sum of dy * xhat over every voxel of the batch that cancels to a small remainder, and an encoder
# tensor sits behind 9-16 batch-statistics backward passes, each of which subtracts two such sums from every element; float32 round-off is
# amplified by the cancellation, and which tensor is worst moves with the summation order of any kernel (stock PyTorch float32 on the CPU is
# 7.0e-3 away from the same golden).  So: 5e-3 for everything else -- decoder / up-conv / head WEIGHTS, the tensors a wrong tap or border class
# shows up in -- 1.2e-2 for the two named classes, and at most a third of all tensors above 5e-3: a 2x regression of either class fails.
LOOSE_GRAD_TENSORS = ("down_tr", ".bn1.weight", ".bn1.bias", ".bn.weight", ".bn.bias")


def _grad_report(model, fx, rel_tol, rel_tol_big, zero_tol, tight=None):
    """rel_tol / rel_tol_big: the gate for small / large tensors; tight: if given, the gate for every tensor OUTSIDE LOOSE_GRAD_TENSORS (which keep
    rel_tol) and the count constraint above."""
    worst, bad, above = 0.0, [], 0
    total = 0
    for name, p in model.named_parameters():
        if f"grad/{name}/none" in fx.files:
            assert p.grad is None, f"{name}: reference has no gradient (unused head), HIP path produced one"
            continue
        assert p.grad is not None, f"{name}: missing gradient"
        ref_s, l2 = fx[f"grad/{name}/samples"], float(fx[f"grad/{name}/l2"])
        got_s = samples(p.grad, 64)
        got_l2 = float(p.grad.double().norm())
        if name.endswith(ZERO_GRAD):
            assert np.abs(got_s).max() <= zero_tol and l2 < 1e-8, (name, np.abs(got_s).max(), l2)
            continue
        rel = np.linalg.norm(got_s - ref_s) / max(np.linalg.norm(ref_s), 1e-30)
        rel_n = abs(got_l2 - l2) / l2
        tol = rel_tol_big if p.numel() > 4096 else rel_tol
        if tight is not None and not any(k in name for k in LOOSE_GRAD_TENSORS):
            tol = tight
        worst = max(worst, rel)
        total += 1
        above += rel > 5e-3
        if rel > tol or rel_n > tol:
            bad.append((name, rel, rel_n))
    assert not bad, f"gradients outside rel-L2 tolerance: {bad[:8]} (+{max(0, len(bad) - 8)} more)"
    if tight is not None:
        assert above * 3 <= total, f"{above} of {total} tensors above 5e-3"
    return worst


def test_fp32_step_matches_reference_golden(golden):
    fx, batches = golden
    model = build(torch.float32)
    r = forward_losses(model, batches[0], int(fx["meta/epoch"]), int(fx["meta/seed"]))
    assert r["index2"] == int(fx["step0/index2"])
    np.testing.assert_allclose(samples(r["mask1"], 256), fx["fwd/out/samples"], rtol=0, atol=5e-5)
    for i in range(3):
        np.testing.assert_allclose(r["dec1"][i][0].detach().double().cpu().numpy(), fx[f"fwd/pro{i}"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(r["dec1"][i][1].detach().double().cpu().numpy(), fx[f"fwd/pre{i}"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(samples(r["mid1"][i], 256), fx[f"fwd/mid{i}/samples"], rtol=0, atol=5e-5)
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        assert abs(float(r[k].detach()) - float(fx[f"step0/{k}"])) < 1e-5, (k, float(r[k]), float(fx[f"step0/{k}"]))
    r["loss"].backward()
    # stock PyTorch float32 (CPU, oneDNN) lands at 7e-3 on this metric: backward through 17 batch-statistics normalisations
    # amplifies float32 round-off, and which tensor is worst moves with the summation order (measured here: 0.6e-2 .. 1.05e-2)
    worst = _grad_report(model, fx, rel_tol=1.2e-2, rel_tol_big=1.2e-2, zero_tol=1e-5, tight=5e-3)
    print(f"fp32: worst gradient rel-L2 vs fp64 golden = {worst:.2e}")
    # BN running statistics after the three forwards of one step
    model.flush_counters()
    sd = model.state_dict()
    for name in sd:
        if O.is_buffer(name):
            np.testing.assert_allclose(sd[name].double().cpu().numpy(), fx[f"buf1/{name}"], rtol=2e-5, atol=2e-6, err_msg=name)


def test_fp32_two_sgd_steps_match_reference_golden(golden):
    fx, batches = golden
    model = build(torch.float32)
    opt = FusedSGD(model.parameters(), lr=float(fx["meta/lr"]), momentum=0.9, weight_decay=1e-4)
    random.seed(int(fx["meta/seed"]))
    for s, batch in enumerate(batches):
        out = train_step(model, opt, batch, int(fx["meta/epoch"]), MSELoss(), CosineSimilarityMean())
        for k, v in zip(("loss", "loss1", "loss2", "loss4", "local_loss"), out):
            tol = 2e-5 if (s == 0 or k in ("loss1", "loss4")) else 5e-4
            d = abs(float(v) - float(fx[f"step{s}/{k}"]))
            print(f"fp32 step {s} {k}: |d|={d:.2e}")
            assert d < tol, (s, k, float(v), float(fx[f"step{s}/{k}"]))
    worst = 0.0
    for name, p in model.named_parameters():
        d = np.abs(samples(p, 64) - fx[f"final/{name}/samples"]).max()
        worst = max(worst, d)
        assert d < 5e-5, (name, d)
    print(f"fp32: worst parameter |d| after 2 SGD steps = {worst:.2e}")


def test_bf16_step_within_stated_tolerance_of_golden(golden):
    """bf16 engine vs the float64 golden step of the real reference (b=4, 32x32x16).  Losses within 3e-2, features by cosine.
    Gradients: the cosine terms reach the decoder through BatchNorm1d over FOUR samples, which amplifies bf16 rounding into
    O(1) changes of the gradients upstream of it (a bf16-emulated run of the ORACLE itself deviates by up to 0.82 rel-L2 there;
    merely re-ordering float32 K-summations -- split-K on/off, a different split count -- moved the norms of the up_tr64 stage
    (which feeds BatchNorm1d(64)) from <0.3 to 0.6 to 4.5 times the golden while everything else stayed put).  Asserted: every
    gradient finite and present/absent as in the reference; MEDIAN norm deviation over the tensors < 0.25 (measured 0.03-0.12);
    at most a quarter of the tensors off by more than 0.3 (measured: 13 of 71, the up_tr64 stage).  `tests/chaos_probe.py` shows
    the spread between equally valid kernel choices on this input: float32 reproduces every golden norm to 1.000 with the brick,
    gather and split-K kernels alike, bfloat16 gives |g|/|g_golden| of up_tr64.up_conv.weight = 5.5 / 0.84 / 0.83 for the three.
    The tight bf16 gradient check is test_restoration_path_gradients_vs_live_oracle (well-conditioned MSE path)."""
    fx, batches = golden
    model = build(torch.bfloat16)
    r = forward_losses(model, batches[0], int(fx["meta/epoch"]), int(fx["meta/seed"]))
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        d = abs(float(r[k].detach()) - float(fx[f"step0/{k}"]))
        print(f"bf16 {k}: {float(r[k].detach()):+.5f} vs {float(fx[f'step0/{k}']):+.5f} |d|={d:.2e}")
        assert d < 3e-2, k
    assert np.abs(samples(r["mask1"], 256) - fx["fwd/out/samples"]).max() < 3e-2
    for i in range(3):
        for j, nm in enumerate(("pro", "pre")):
            a, b = r["dec1"][i][j].detach().double().cpu().numpy().ravel(), fx[f"fwd/{nm}{i}"].ravel()
            cs = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
            print(f"bf16 {nm}{i}: cosine to golden {cs:.5f}")
            assert cs > 0.98, (nm, i, cs)
    r["loss"].backward()
    devs = []
    for name, p in model.named_parameters():
        if f"grad/{name}/none" in fx.files:
            assert p.grad is None, name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        if name.endswith(ZERO_GRAD):
            continue
        l2 = float(fx[f"grad/{name}/l2"])
        devs.append((abs(float(p.grad.double().norm()) - l2) / l2, name))
    devs.sort(reverse=True)
    print("bf16: gradient-norm deviation vs fp64 golden, worst five:", [(round(d, 3), n) for d, n in devs[:5]],
          "median", round(devs[len(devs) // 2][0], 3))
    far = [(round(d, 2), n) for d, n in devs if d > 0.3]
    print("bf16: deviations > 0.3:", far)
    assert devs[len(devs) // 2][0] < 0.25, devs[len(devs) // 2]
    assert len(far) <= 0.25 * len(devs), far


# Well-conditioned and BASELINE-shaped golden steps of the REAL reference (float64): BatchNorm1d over 16 rows (c_b16_32x32x16) instead of
# the 4 of c_small_b4, and the BASELINE crop size 64x64x32 (c_luna_b8 at b = 8: b = 16 at this size needs > 63 GB in float64 in the
# authoring container; c_luna_b2 at b = 2, where BatchNorm1d over TWO rows makes the cosine path ill-conditioned: forward maps and the
# MSE terms are held tight there, the cosine terms loosely, no gradients).
# Measured on MI355X (c_b16): float32 maps 5e-6, losses 2e-7, worst gradient rel-L2 8.5e-3 (stock torch float32: 7e-3);
# bfloat16 maps max 1.4e-2 (3.8e-2 on the full-resolution deep-supervision map) / mean 3e-3..8e-3, feature cosine >= 0.9983,
# MSE losses 2e-7 / 5e-6, cosine losses 8e-4 / 3e-4, gradient norms median 0.9 % / worst 16 % off, gradient direction >= 0.87 --
# the tolerances below leave 1.5-2x on those; the b = 4 fixture needed 3e-2 on the losses and 25 % / 30 % on the gradient norms.
GOLDEN_STEPS = {
    # fp32: (map max, feature abs, (MSE-loss, other-loss) abs, gradient rel-L2)
    # bf16: (map max [mean = 1/4 of it], feature cosine, (MSE-loss, other-loss) abs, gradient-norm median, gradient-norm worst, gradient cosine min)
    "c_b16_32x32x16": dict(tight=5e-3, fp32=(5e-5, 2e-4, (1e-5, 1e-5), 1.2e-2), bf16=(6e-2, 0.997, (5e-5, 2e-3), 0.02, 0.25, 0.8), grads=True),
    # b = 8 rows in BatchNorm1d, 8x the voxels per crop: measured bf16 maps max 5.8e-2, cosine losses 1.5e-3, norms median 1.8 % / worst 10 %, direction 0.85
    # (tight: the decoder weights' gate; 8x the voxels per sum at the BASELINE crop size: measured worst up_tr128.ops.1.conv1.weight 5.4e-3)
    "c_luna_b8_64x64x32": dict(tight=7e-3, fp32=(5e-5, 2e-4, (1e-5, 1e-5), 1.2e-2), bf16=(9e-2, 0.996, (5e-5, 4e-3), 0.04, 0.25, 0.75), grads=True),
    "c_luna_b2_64x64x32": dict(fp32=(5e-5, 5e-4, (1e-5, 2e-5), None), bf16=(1e-1, 0.975, (5e-5, 2e-2), None, None, None), grads=False),
}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", list(GOLDEN_STEPS))
def test_step_matches_reference_golden_large_batch(tag, dt, golden_dir):
    path = os.path.join(golden_dir, tag + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{tag}.npz not generated")
    fx = np.load(path)
    spec = GOLDEN_STEPS[tag]
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=7)
    model = build(dt)
    r = forward_losses(model, batch, int(fx["meta/epoch"]), int(fx["meta/seed"]))
    assert r["index2"] == int(fx["step0/index2"])
    f32 = dt == torch.float32
    tol = spec["fp32"] if f32 else spec["bf16"]
    dm = [np.abs(samples(r["mask1"], 256) - fx["fwd/out/samples"])] + [np.abs(samples(r["mid1"][i], 256) - fx[f"fwd/mid{i}/samples"]) for i in range(3)]
    d_map, d_mean = max(d.max() for d in dm), max(d.mean() for d in dm)
    print(f"{tag} {dt}: sigmoid maps (out, 3 deep-supervision maps) max|d| = {[float('%.2e' % d.max()) for d in dm]} mean|d| = {[float('%.2e' % d.mean()) for d in dm]}")
    assert d_map < tol[0] and d_mean < tol[0] / 4
    for i in range(3):
        for j, nm in enumerate(("pro", "pre")):
            a, ref = r["dec1"][i][j].detach().double().cpu().numpy(), fx[f"fwd/{nm}{i}"]
            if f32:
                assert np.abs(a - ref).max() < tol[1], (nm, i, np.abs(a - ref).max())
            else:
                cs = float(a.ravel() @ ref.ravel() / (np.linalg.norm(a) * np.linalg.norm(ref)))
                print(f"{tag} bf16 {nm}{i}: cosine to golden {cs:.5f}")
                assert cs > tol[1], (nm, i, cs)
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        d = abs(float(r[k].detach()) - float(fx[f"step0/{k}"]))
        print(f"{tag} {dt} {k}: {float(r[k].detach()):+.6f} vs {float(fx[f'step0/{k}']):+.6f} |d|={d:.2e}")
        assert d < tol[2][0 if k in ("loss1", "loss4") else 1], (k, d)
    if not spec["grads"]:
        return
    r["loss"].backward()
    if f32:
        worst = _grad_report(model, fx, rel_tol=tol[3], rel_tol_big=tol[3], zero_tol=1e-5, tight=spec.get("tight", 5e-3))
        print(f"{tag} fp32: worst gradient rel-L2 vs fp64 golden = {worst:.2e}")
        return
    devs, coss, scalars = [], [], []
    for name, p in model.named_parameters():
        if f"grad/{name}/none" in fx.files:
            assert p.grad is None, name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        if name.endswith(ZERO_GRAD):
            continue
        l2 = float(fx[f"grad/{name}/l2"])
        (scalars if p.numel() == 1 else devs).append((abs(float(p.grad.double().norm()) - l2) / l2, name))
        gs, rs = samples(p.grad, 64), fx[f"grad/{name}/samples"]
        if p.numel() >= 64:
            coss.append((float(gs @ rs / max(np.linalg.norm(gs) * np.linalg.norm(rs), 1e-30)), name))
    devs.sort(reverse=True)
    coss.sort()
    print(f"{tag} bf16: gradient-norm deviation worst five {[(round(d, 3), n) for d, n in devs[:5]]} median {devs[len(devs) // 2][0]:.3f}")
    print(f"{tag} bf16: gradient direction (cosine on 64 samples) worst five {[(round(c, 3), n) for c, n in coss[:5]]}")
    assert devs[len(devs) // 2][0] < tol[3] and devs[0][0] < tol[4], devs[:3]
    # One-element parameters (weight / bias of the deep-supervision heads' 1-channel BatchNorm): the "norm" is one signed sum over every voxel
    # of the batch of terms of both signs -- its relative error is set by cancellation, not by the kernels: 0.20 with up_tr128's convolutions
    # on the 4x8x8-brick kernel, 0.28 on the wide-brick kernel (another summation order; the median over all parameters went 0.011 -> 0.008).
    # The float32 mode pins the same sums to 1e-2.
    scalars.sort(reverse=True)
    print(f"{tag} bf16: one-element parameters {[(round(d, 3), n) for d, n in scalars]}")
    assert all(d < 0.4 for d, _ in scalars), scalars
    assert coss[0][0] > tol[5], coss[:3]


def test_fused_cosine_terms_equal_the_26_separate_launches():
    """train_3d.step_losses hands the 13 cos_loss calls of a step (26 cosine means) to ONE kernel launch (pcrl_cosine_terms_*); with
    train_3d.FUSED_COS_LOSSES = False it calls the cosine kernel 26 times like the reference calls nn.CosineSimilarity.  Same draws
    from python's `random`, same losses (float32 summation order aside) and same gradients."""
    from pcrlv2_amd import train_3d as T
    batch = O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=11)
    res = []
    for fused in (True, False):
        T.FUSED_COS_LOSSES = fused
        try:
            model = build(torch.float32)
            random.seed(5)
            losses = T.step_losses(model, batch, 3, MSELoss(), CosineSimilarityMean())
            after = random.random()
            losses[0].backward()
            res.append(([float(l) for l in losses], {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}, after))
        finally:
            T.FUSED_COS_LOSSES = True
    (la, ga, ra), (lb, gb, rb) = res
    assert ra == rb                                                   # the same 13 draws were consumed
    for a, b in zip(la, lb):
        assert abs(a - b) < 2e-6, (la, lb)
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n
        if ga[n] is not None:
            d = float((ga[n] - gb[n]).norm()) / max(float(gb[n].norm()), 1e-12)
            assert d < 1e-4 or float(gb[n].abs().max()) < 1e-7, (n, d)


def test_loss_tail_equals_the_scalar_expression_and_its_autograd():
    """functions.loss_tail (pcrl_loss_total + pcrl_loss_total_bwd, the [2] cosine vector taken whole) against the reference's scalar
    expression loss1 + loss2 + beta * l4 + local_loss (train_3d.py:136-138) built from torch operators: same values, same gradients for
    loss1, l4 and both cosine groups; a root gradient other than 1 scales them."""
    from pcrlv2_amd import functions as F_
    vals = torch.tensor([0.37, -0.81, 1.93, -0.42], device=DEV)
    beta = 0.731
    for root in (None, 2.5):
        l1, l4 = vals[0].clone().requires_grad_(True), vals[2].clone().requires_grad_(True)
        cos = torch.stack([vals[1], vals[3]]).clone().requires_grad_(True)
        total, scaled, lg, ll = F_.loss_tail(l1, cos, l4, beta)
        r1, r4, rc = vals[0].clone().requires_grad_(True), vals[2].clone().requires_grad_(True), torch.stack([vals[1], vals[3]]).clone().requires_grad_(True)
        ref = r1 + rc[0] + beta * r4 + rc[1]
        assert abs(float(total) - float(ref)) < 1e-6 and abs(float(scaled) - beta * float(vals[2])) < 1e-6
        assert float(lg) == float(vals[1]) and float(ll) == float(vals[3])
        assert not scaled.requires_grad and not lg.requires_grad and not ll.requires_grad
        if root is None:
            total.backward(gradient=F_.root_gradient(total))
            ref.backward()
        else:
            total.backward(gradient=torch.tensor(root, device=DEV))
            ref.backward(gradient=torch.tensor(root, device=DEV))
        for a, b in ((l1.grad, r1.grad), (l4.grad, r4.grad), (cos.grad, rc.grad)):
            assert torch.allclose(a, b, rtol=1e-6, atol=0), (a, b)


@pytest.mark.parametrize("switch", ["COMPOSE_UPCONV", "FOLD_POOL_GRAD", "FOLD_GAP_GRAD", "FUSE_APPLY_CONSUMERS"])
def test_round2_fusions_leave_the_step_unchanged_fp32(switch):
    """Each memory-pass fusion of round 2 (config.py) against the separate kernels it replaces, on a whole training step in the exact-fp32
    mode: same losses, same gradient for every parameter (float32 summation order -- and, for the composed up-conv, the association of
    two nested sums -- aside).  COMPOSE_UPCONV: UpTransition's ConvTranspose3d -> Conv3d as one operator, gradients of up_conv.weight /
    up_conv.bias / ops.0.conv1.weight through the composed weights, accumulated over the three passes and delivered once per backward()."""
    from pcrlv2_amd import config, train_3d as T
    batch = O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=12)
    res = []
    for on in (True, False):
        old = getattr(config, switch)
        setattr(config, switch, on)
        try:
            model = build(torch.float32)
            random.seed(7)
            T.begin_step()
            losses = T.step_losses(model, batch, 3, MSELoss(), CosineSimilarityMean())
            losses[0].backward()
            res.append(([float(l) for l in losses], {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}))
        finally:
            setattr(config, switch, old)
    (la, ga), (lb, gb) = res
    for a, b in zip(la, lb):
        assert abs(a - b) < 5e-6, (switch, la, lb)
    worst = ("", 0.0)
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), (switch, n)
        if ga[n] is not None:
            d = float((ga[n] - gb[n]).norm()) / max(float(gb[n].norm()), 1e-12)
            if float(gb[n].abs().max()) >= 1e-7 and d > worst[1]:
                worst = (n, d)
    print(f"  {switch}: worst relative gradient difference {worst[1]:.2e} ({worst[0]})")
    # The three folds change no arithmetic (measured 0 .. 2e-6).  The composed up-conv changes the association of two nested float32 sums:
    # y0 moves by ~1e-6 relative, and the backward through 17 batch-statistics normalisations and the cosine terms amplifies float32
    # round-off to ~1e-2 against the float64 golden on EITHER route (test_fp32_step_matches_reference_golden: 0.6e-2 .. 1.05e-2; stock
    # PyTorch float32: 7e-3) -- the two routes differ from each other by 4e-4 .. 4e-3, uniformly over all layers.
    assert worst[1] < (1e-2 if switch == "COMPOSE_UPCONV" else 2e-5), (switch, worst)


@pytest.mark.parametrize("switch", ["WGRAD_SIDE_STREAM_3D", "FWD_BRANCH_STREAM", "VIEW_STREAMS", "EARLY_COMPOSED", "VIEW_WGRAD_INLINE", "PREPACK", "FUSED_GRAD_SUM"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_weight_gradients_on_the_side_stream_are_bit_identical(dt, switch):
    """config.WGRAD_SIDE_STREAM_3D (default on): the weight-gradient kernels run on a second stream next to the data-gradient / BatchNorm
    chain; config.FWD_BRANCH_STREAM (default on): the decoder stages' heads and deep-supervision maps run there in forward.  Same kernels,
    same operands, only the ordering between streams differs -> the parameters after two SGD steps and the BatchNorm running statistics
    must be BIT-identical to the one-stream run."""
    from pcrlv2_amd import config
    batches = [O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=21 + s) for s in range(2)]
    finals = []
    old = getattr(config, switch)
    try:
        for on in (True, False):
            setattr(config, switch, on)
            model = build(dt)
            opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
            random.seed(5)
            for bt in batches:
                losses = train_step(model, opt, bt, 3, MSELoss(), CosineSimilarityMean())
            torch.cuda.synchronize()
            finals.append(([float(l) for l in losses], opt.flat_p.clone(), {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    finally:
        setattr(config, switch, old)
    (la, pa, ra), (lb, pb, rb) = finals
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb), float((pa - pb).abs().max())
    for k in ra:
        assert torch.equal(ra[k], rb[k]), k


def test_three_stream_step_is_bit_identical_at_the_bench_size():
    """The same check at BASELINE's size (b = 32, 64x64x32 + 6 local views: other launch timing, every kernel at full occupancy): four steps
    with the second view's stream, the side branches and the side-stream weight gradients all on, twice, against the one-stream run --
    parameters, momentum buffers and running statistics bit for bit (tools/stream_stress.py is the longer form)."""
    from bench import synthetic_batch
    from pcrlv2_amd import config
    batch = synthetic_batch(32, (64, 64, 32), 16, torch.device(DEV), 7)
    keep = (config.WGRAD_SIDE_STREAM_3D, config.FWD_BRANCH_STREAM, config.VIEW_STREAMS)

    def run(on):
        config.WGRAD_SIDE_STREAM_3D = config.FWD_BRANCH_STREAM = config.VIEW_STREAMS = on
        torch.manual_seed(0)
        random.seed(0)
        model = PCRLv23d().to(DEV).train().set_compute_dtype(torch.bfloat16)
        opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
        for _ in range(4):
            out = train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean(), guard=False)
        torch.cuda.synchronize()
        rs = torch.cat([v.flatten().float() for k, v in sorted(model.state_dict().items()) if "running" in k])
        return [float(o) for o in out], opt.flat_p.clone(), opt.flat_buf.clone(), rs

    try:
        ref = run(False)
        for _ in range(2):
            got = run(True)
            assert got[0] == ref[0], (got[0], ref[0])
            for a, b, what in zip(got[1:], ref[1:], ("parameters", "momentum buffers", "running statistics")):
                assert torch.equal(a, b), what
    finally:
        config.WGRAD_SIDE_STREAM_3D, config.FWD_BRANCH_STREAM, config.VIEW_STREAMS = keep


def test_optional_groupnorm_silu_mode_vs_torch_definition():
    """PCRLv23d(norm='gn', act='silu') -- an OPTIONAL, NON-REFERENCE mode (BASELINE.json's north_star names GroupNorm + SiLU; the
    reference's own norm='gn' crashes at construction and it rejects 'silu').  Checked against the oracle's torch
    restatement (F.group_norm / F.silu in float64) in float32: outputs and every parameter gradient of a restoration + feature loss."""
    torch.manual_seed(3)
    model = PCRLv23d(norm="gn", act="silu").to(DEV).train()
    model.set_compute_dtype(torch.float32)
    sd = model.state_dict()
    assert "down_tr64.ops.0.bn1.running_mean" not in sd and "up_tr64.deep_supervision_head.bn1.running_mean" in sd
    b, dhw = 2, (16, 16, 16)
    x, _, gt, _, _ = O.fill_batch(b, dhw, dtype=torch.float32, seed=5)
    out, feats, masks = model(x.to(DEV))
    loss = ((out - gt.to(DEV)) ** 2).mean() + sum(f.sum() * 0.01 for pair in feats for f in pair) + masks[0].mean()
    loss.backward()
    st = {k: (v.detach().double().cpu().requires_grad_(v.is_floating_point() and not O.is_buffer(k)) if v.is_floating_point() else v.cpu())
          for k, v in sd.items()}
    with torch.backends.mkldnn.flags(enabled=False):
        o_out, o_feats, o_masks = O.forward(st, x.double())
        o_loss = ((o_out - gt.double()) ** 2).mean() + sum(f.sum() * 0.01 for pair in o_feats for f in pair) + o_masks[0].mean()
        o_loss.backward()
    assert abs(float(loss) - float(o_loss)) < 2e-5
    np.testing.assert_allclose(out.detach().double().cpu().numpy(), o_out.detach().numpy(), rtol=0, atol=5e-5)
    worst = 0.0
    for name, p in model.named_parameters():
        g, r = p.grad, st[name].grad
        assert (g is None) == (r is None), name
        if g is None:
            continue
        rel = float((g.double().cpu() - r).norm() / max(float(r.norm()), 1e-12))
        if float(r.norm()) < 1e-9:   # biases in front of a BATCH norm (the deep-supervision heads keep BatchNorm): exact zero
            assert float(g.abs().max()) < 1e-5, name
            continue
        worst = max(worst, rel)
        assert rel < 2e-2, (name, rel)
    print(f"gn+silu mode: worst gradient rel-L2 vs float64 torch definition = {worst:.2e}")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_eval_mode_forward_matches_reference_golden(dt, golden_dir):
    """model.eval() on the HIP path against the REAL reference in .eval() (tests/golden/eval_b2_32x32x16.npz).  The state (running
    statistics moved by one training step) is rebuilt with the deterministic oracle.  float32: maps 5e-5 / features 2e-4 abs (the
    forward envelope of SURVEY App. C); bfloat16: maps 2e-2 abs, features by cosine > 0.995."""
    fx = np.load(os.path.join(golden_dir, "eval_b2_32x32x16.npz"))
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    with torch.backends.mkldnn.flags(enabled=False):
        st1, _, _, _ = O.train_steps(O.fill_state(torch.float64), [O.fill_batch(b, dhw, dtype=torch.float64, seed=int(fx["meta/state_batch_seed"]))], 0, 1e-3, 240, 0)
    x = O.fill_batch(b, dhw, dtype=torch.float32, seed=int(fx["meta/input_seed"]))[0].to(DEV)
    model = build(dt, {k: (v.float() if v.is_floating_point() else v) for k, v in st1.items()})
    model.eval()
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}
    out, feats, masks = model(x)
    assert not out.requires_grad and len(masks) == 3 and out.shape == x.shape
    map_tol = 5e-5 if dt == torch.float32 else 2e-2
    assert np.abs(samples(out, 512) - fx["out/samples"]).max() < map_tol
    for i in range(3):
        assert np.abs(samples(masks[i], 512) - fx[f"mask{i}/samples"]).max() < map_tol, i
        for j, nm in enumerate(("pro", "pre")):
            a, r = feats[i][j].double().cpu().numpy(), fx[f"{nm}{i}"]
            if dt == torch.float32:
                np.testing.assert_allclose(a, r, rtol=0, atol=2e-4, err_msg=f"{nm}{i}")
            else:
                cs = float(a.ravel() @ r.ravel() / (np.linalg.norm(a) * np.linalg.norm(r)))
                assert cs > 0.995, (nm, i, cs)
    # inference leaves parameters, running statistics and counters alone; local=True returns no masks
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd_before[k]), k
    assert model(x, local=True)[2] == []
    model.train()
    assert model(x)[0].requires_grad


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_restoration_path_gradients_vs_live_oracle(dt):
    """loss = MSE(out, gt) + MSE(mid[2], gt) + MSE(mid[0], gt) on one view (every conv / BN / pool / convT / trilinear
    kernel is on this path) against the float64 oracle run live.  Backward through 17 batch-statistics normalisations
    amplifies rounding (float32 itself lands at ~1e-2 on the first layers), so: float32 per-tensor rel-L2 < 2e-2;
    bfloat16: direction agreement -- cosine(g_bf16, g_fp64) > 0.7 on every weight tensor and norm ratio within 25 %."""
    import torch.nn.functional as F
    b, dhw = 4, (32, 32, 16)
    st32 = O.fill_state(torch.float32)
    x, _, gt, _, _ = O.fill_batch(b, dhw, dtype=torch.float32, seed=11)
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in st32.items()}
    pn = [k for k in st64 if not O.is_buffer(k)]
    for k in pn:
        st64[k].requires_grad_(True)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    with torch.backends.mkldnn.flags(enabled=False):
        out, _, mid = O.forward(st64, x.double())
        lref = F.mse_loss(out, gt.double()) + F.mse_loss(mid[2], gt.double()) + F.mse_loss(mid[0], gt.double())
        gref = dict(zip(pn, torch.autograd.grad(lref, [st64[k] for k in pn], allow_unused=True)))
    model = build(dt, st32)
    crit = MSELoss()
    o, _, m = model(x.to(DEV))
    gtd = gt.to(DEV)
    loss = crit(o, gtd) + crit(m[2], gtd) + crit(m[0], gtd)
    assert abs(float(loss.detach()) - float(lref.detach())) < (1e-5 if dt == torch.float32 else 2e-3)
    loss.backward()
    rows = []
    for name, p in model.named_parameters():
        g = gref[name]
        if g is None:
            assert p.grad is None, name
            continue
        if name.endswith(ZERO_GRAD):
            continue
        a = p.grad.double().cpu().reshape(-1)
        g = g.reshape(-1)
        rows.append((name, float((a - g).norm() / g.norm()), float(a @ g / (a.norm() * g.norm())), float(a.norm() / g.norm()), p.numel()))
    for name, rel, cs, nr, n in rows:
        print(f"  {dt} {name:45s} rel-L2 {rel:.3e} cos {cs:.4f} norm-ratio {nr:.3f}")
    if dt == torch.float32:
        assert max(r[1] for r in rows) < 2e-2, max(rows, key=lambda r: r[1])
    else:
        big = [r for r in rows if r[4] >= 1024]
        assert min(r[2] for r in big) > 0.7, min(big, key=lambda r: r[2])
        assert all(0.75 < r[3] < 1.25 for r in big), [r for r in big if not 0.75 < r[3] < 1.25]


def test_fp32_matches_live_oracle_other_inputs():
    """Oracle run here on CPU (float64) on inputs the fixtures do not cover: b=5 (odd batch), 16x24x8 volumes."""
    b, dhw = 5, (16, 24, 8)
    st32 = O.fill_state(torch.float32)
    batch = O.fill_batch(b, dhw, local=8, dtype=torch.float32, seed=42)
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in st32.items()}
    b64 = tuple(t.double() if torch.is_tensor(t) else [u.double() for u in t] for t in batch)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    pn = [k for k in st64 if not O.is_buffer(k)]
    for k in pn:
        st64[k].requires_grad_(True)
    nb = {}
    with torch.backends.mkldnn.flags(enabled=False):
        ref = O.step_losses(st64, b64, 7, random.Random(3), nb)
        gref = dict(zip(pn, torch.autograd.grad(ref["loss"], [st64[k] for k in pn], allow_unused=True)))
    model = build(torch.float32, st32)
    r = forward_losses(model, batch, 7, 3)
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        assert abs(float(r[k]) - float(ref[k])) < 1e-5, k
    assert (r["mask1"].double().cpu() - ref["mask1"].detach()).abs().max() < 5e-5
    r["loss"].backward()
    for name, p in model.named_parameters():
        g = gref[name]
        if g is None:
            assert p.grad is None, name
            continue
        if name.endswith(ZERO_GRAD):
            assert p.grad.abs().max() < 1e-5, name
            continue
        rel = float((p.grad.double().cpu() - g).norm() / g.norm())
        assert rel < 1e-2, (name, rel)
    model.flush_counters()
    sd = model.state_dict()
    for k, v in nb.items():
        np.testing.assert_allclose(sd[k].double().cpu().numpy(), v.double().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("layer", [(32, 64, 64, 32, 128, 64), (32, 16, 16, 8, 512, 256)])
def test_full_size_conv_adjoint_identities_bf16(layer):
    """BASELINE C2 sizes (b=32, bf16).  No CPU reference is affordable here, but the three conv kernels must be
    mutually adjoint:  <conv(x; w), dy> == <x, dgrad(dy; w)> == <w, wgrad(x, dy)>  (size-independent property)."""
    N, D, H, W, Ci, Co = layer
    dt = torch.bfloat16
    L, s = lib(), stream_handle()
    g = torch.Generator(device=DEV).manual_seed(1)
    x = ops.new_act(N, D, H, W, Ci, dt, DEV)
    dy = ops.new_act(N, D, H, W, Co, dt, DEV)
    x.normal_(generator=g)
    dy.normal_(generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, device=DEV, generator=g) * 0.05
    pk = ops.PackedWeights("conv3")
    wf, wd = pk.get(w, dt)
    y = ops.new_act(N, D, H, W, Co, dt, DEV)
    L.call("pcrl_conv3d_k3_fwd", x, wf, None, y, None, N, D, H, W, Ci, Co, dtype_code(dt), s)
    # dy correlated with y (cosine ~0.9): an independent random dy makes every inner product ~1e-4 of |y||dy| -- the identities would then
    # hold to the tolerance below whatever the kernels computed
    dy = ((y.float() / y.float().std()) + 0.5 * dy.float()).to(dt)
    dx = ops.new_act(N, D, H, W, Ci, dt, DEV)
    L.call("pcrl_conv3d_k3_fwd", dy, wd, None, dx, None, N, D, H, W, Co, Ci, dtype_code(dt), s)
    nb = L.call("pcrl_conv3d_k3_wgrad_ws_bytes", N, D, H, W, Ci, Co)
    dw = torch.zeros_like(w)
    L.call("pcrl_conv3d_k3_wgrad", x, dy, dw, ops.workspace(nb, x.device), nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
    wq = w.to(dt).double()
    a = float((y.double() * dy.double()).sum())
    b_ = float((x.double() * dx.double()).sum())
    c = float((wq * dw.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    print(f"adjoint: <y,dy>={a:.6e} <x,dx>={b_:.6e} <w,dw>={c:.6e} (|y||dy|={scale:.3e})")
    # bf16 output rounding is unbiased noise: inner products agree to ~2^-9/sqrt(#elements) of the norm product
    assert abs(a) > 0.5 * scale
    assert abs(a - c) < 2e-4 * scale and abs(b_ - c) < 2e-4 * scale


def test_full_size_step_properties_bf16():
    """One BASELINE-C2 step (b=32, 64x64x32 + 6x16^3, bf16): finite losses in the expected ranges, every used
    parameter receives a finite gradient, the three unused deep-supervision heads receive none, BN running
    statistics moved, and the step is deterministic (bit-identical when repeated from the same state)."""
    torch.manual_seed(0)
    b = 32
    gen = torch.Generator().manual_seed(1234)
    x1 = torch.randn(b, 1, 64, 64, 32, generator=gen)
    batch = (x1, x1 + 0.1 * torch.randn(b, 1, 64, 64, 32, generator=gen), torch.rand(b, 1, 64, 64, 32, generator=gen), None,
             [torch.randn(b, 1, 16, 16, 16, generator=gen) for _ in range(6)])
    results = []
    for rep in range(2):
        torch.manual_seed(0)
        model = PCRLv23d().to(DEV).train().set_compute_dtype(torch.bfloat16)
        opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        random.seed(0)
        out = train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean())
        vals = [float(v) for v in out]
        assert all(np.isfinite(vals)), vals
        loss, loss1, loss2, loss4, local = vals
        assert 0.0 < loss1 < 0.5 and -1.0 <= loss2 <= 1.0 and -1.0 <= local <= 1.0 and 0.0 < loss4 < 0.5
        n_none = sum(p.grad is None for p in model.parameters())
        assert n_none == 8
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        sd = model.state_dict()
        assert float(sd["up_tr64.ops.0.bn1.running_var"].sub(1).abs().max()) > 1e-4
        assert int(sd["down_tr64.ops.0.bn1.num_batches_tracked"]) == 3
        results.append((vals, opt.flat_p.clone()))
    assert results[0][0] == results[1][0], "losses differ between identical runs (non-deterministic reduction?)"
    assert torch.equal(results[0][1], results[1][1]), "parameters differ between identical runs"


def test_data_parallel_wrapper_on_one_rank_rccl_group():
    """ddp.DataParallel on a 1-rank RCCL group: the hook-driven bucket launches (final-pass marks set by the stage
    Functions during view 1's backward), the side stream and the collectives run for real; with world = 1 the result must
    be bit-identical to the plain step.  (Multi-GPU runs are the driver's; the N>1 logic is covered with gloo on CPU.)"""
    import torch.distributed as dist
    from pcrlv2_amd import ddp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
    b, dhw = 4, (32, 32, 16)
    batches = [O.fill_batch(b, dhw, dtype=torch.float32, seed=50 + s) for s in range(2)]
    finals = []
    for wrap in (False, True):
        model = build(torch.bfloat16)
        opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
        launched_in_backward = []
        if wrap:
            dp = ddp.DataParallel(model, opt, bucket_mb=8.0, overlap=True, force_collectives=True)
            assert len(dp.reducer.buckets) >= 4
            orig = dp._pre_step

            def spy(o, h, dp=dp, orig=orig):
                launched_in_backward.append(sum(dp._launched))
                return orig(o, h)
            opt.pre_step = spy
        random.seed(1)
        for batch in batches:
            train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean())
        torch.cuda.synchronize()
        finals.append(opt.flat_p.clone())
        if wrap:
            assert not dp._late
            assert all(n >= 1 for n in launched_in_backward), f"no bucket was launched during backward: {launched_in_backward}"
            print("buckets launched before optimizer.step():", launched_in_backward, "of", len(dp.reducer.buckets))
    assert torch.equal(finals[0], finals[1])
    dist.destroy_process_group()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_loss_curve_12_steps_vs_reference_golden(dt, golden_dir):
    """north_star: "loss curve matching reference to 1e-3".  Scoped as SURVEY App. C requires: 12 SGD steps from the same
    state on correlated synthetic views (b=8, 32x32x16) against the curve of the REAL reference in float64
    (tests/golden/curve_b8_32x32x16_12steps.npz, oracle/make_golden.py:make_curve).
      * restoration loss `loss1` (MSE of the full-resolution output): every one of the 12 steps within 1e-3, fp32 and bf16
        (measured: <= 4e-4);
      * deep-supervision loss `loss4`: within 1e-3 for the first 8 steps, 4e-3 through step 11 (float32 and bfloat16 deviate
        by the SAME amount there: it is the parameter trajectory that drifts, driven by the cosine terms, not precision);
      * total loss: float32 within 1e-3 on steps 0-2 and 2e-3 on step 3 (stock PyTorch float32 is 7e-4 away from its own
        float64 run at step 3 and 3e-3 at step 4; this engine measured 0.6e-3 / 1.0e-3 there depending only on whether the 26 cosine
        means are summed by one kernel or by 26); bfloat16 within 5e-3 on steps 0-1 and 2.5e-2 on steps 2-3 (measured 1.6e-2 with the two up-stage convolutions separate, 1.8e-2 composed; the global
        cosine term is already rounding-order dependent there: changing only the summation order of the BatchNorm backward
        partials, or of one bias gradient, moved it between 1e-4 and 7e-3); afterwards the cosine terms diverge chaotically (stock PyTorch float32 does
        too, App. C), so the 12-step MEAN is asserted: 1e-2 (fp32), 4e-2 (bf16; per-step
        deviations of the cosine terms reach 8e-2 from step 5 on and change sign with the summation order of any kernel, so
        the mean of 12 has a spread of ~2e-2: measured 0.6e-2 .. 2.2e-2 over this round's kernel versions)."""
    fx = np.load(os.path.join(golden_dir, "curve_b8_32x32x16_12steps.npz"))
    ref, b, dhw, nsteps = fx["curve"], int(fx["b"]), tuple(int(v) for v in fx["dhw"]), int(fx["nsteps"])
    batches = [O.fill_batch(b, dhw, dtype=torch.float32, seed=int(fx["batch_seed0"]) + s) for s in range(nsteps)]
    model = build(dt)
    opt = FusedSGD(model.parameters(), lr=float(fx["base_lr"]), momentum=0.9, weight_decay=1e-4)
    random.seed(int(fx["seed"]))
    got = []
    for bt in batches:
        out = train_step(model, opt, bt, int(fx["epoch"]), MSELoss(), CosineSimilarityMean())
        got.append([float(v) for v in out])
    names = ("loss", "loss1", "loss2", "loss4", "local_loss")
    for s in range(nsteps):
        print(f"  {dt} step {s:2d}: " + "  ".join(f"{n} {got[s][i]:+.5f} ({got[s][i] - ref[s][i]:+.1e})" for i, n in enumerate(names)))
    for s in range(nsteps):
        assert abs(got[s][1] - ref[s][1]) < 1e-3, (s, "loss1")
        assert abs(got[s][3] - ref[s][3]) < (1e-3 if s < 8 else 4e-3), (s, "loss4")
    for s in range(4):
        tol = (1e-3 if s < 3 else 2e-3) if dt == torch.float32 else (5e-3 if s < 2 else 2.5e-2)
        assert abs(got[s][0] - ref[s][0]) < tol, (s, "loss")
    mean_d = abs(np.mean([g[0] for g in got]) - ref[:, 0].mean())
    assert mean_d < (1e-2 if dt == torch.float32 else 4e-2), mean_d


LONG_CURVE = {}      # dtype -> the engine's 300-step curve (computed once per session: both parametrisations of the test below print and use it)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_long_horizon_loss_curve_300_steps_vs_reference(dt, golden_dir):
    """Row LC (north_star: "loss curve matching reference to 1e-3"; SURVEY App. C iv): 300 SGD steps (b = 8, 32x32x16 + 6 local 16^3, correlated
    synthetic views, lr 1e-3) of the float32 AND the bfloat16 engine against the curve of the REAL reference in FLOAT64 (oneDNN off) over the same
    steps, same state, same draws (tests/golden/lc_b8_32x32x16_300steps.npz, oracle/make_golden.py --long-curve, train_3d.py:109-151).  The fixture also
    holds the reference's own STOCK FLOAT32 run (oneDNN on: its CPU path) and how far THAT drifts from its float64 run -- `stock_fp32_max_abs` per
    component, `stock_fp32_ema_max_after20` -- which is the yardstick: the total swings by +-0.5 from step to step (13 cosine terms through
    BatchNorm1d over eight rows), two float32 runs one ulp apart drift as far as bf16 does (profiles/r05bg_long_run_compare.txt).
    Held, per engine dtype:
      * restoration loss `loss1`: within 1e-3 of the float64 reference at EVERY step (north_star's bound, absolute);
      * deep-supervision loss `loss4` (max), EMA(0.9) of the total after step 20 (max and mean of |difference|): within LC_FACTOR x the drift of
        the reference's own float32 runs (stock oneDNN; oneDNN off -- two draws, the larger counts) from its float64 run on the same quantity:
        no free-standing tolerance.
    Measured values are printed (profiles/r06_long_curve.txt)."""
    fx = np.load(os.path.join(golden_dir, "lc_b8_32x32x16_300steps.npz"), allow_pickle=True)
    assert str(fx["reference_dtype"]).startswith("float64"), "the fixture must hold the float64 reference curve (oracle/make_golden.py --long-curve)"
    ref, b, dhw, nsteps, ema = fx["curve"], int(fx["b"]), tuple(int(v) for v in fx["dhw"]), int(fx["nsteps"]), float(fx["ema"])
    c32 = fx["stock_fp32_curve"][:nsteps]
    model = build(dt)
    opt = FusedSGD(model.parameters(), lr=float(fx["base_lr"]), momentum=0.9, weight_decay=1e-4)
    random.seed(int(fx["seed"]))
    got = []
    for s in range(nsteps):
        bt = O.fill_batch(b, dhw, dtype=torch.float32, seed=int(fx["batch_seed0"]) + s)
        out = train_step(model, opt, bt, int(fx["epoch"]), MSELoss(), CosineSimilarityMean())
        got.append([float(v) for v in out])
    got = np.array(got)
    LONG_CURVE[dt] = got

    def ema_of(v):
        out, e = [], v[0]
        for x in v:
            e = ema * e + (1.0 - ema) * x
            out.append(e)
        return np.array(out)

    def drift(c, base=None):      # of curve c from `base` (default: the float64 reference): loss1 max, loss4 max, EMA(total) after step 20: max and mean of |difference|
        base = ref if base is None else base
        de = np.abs(ema_of(c[:, 0]) - ema_of(base[:, 0]))[20:]
        return np.abs(c[:, 1] - base[:, 1]).max(), np.abs(c[:, 3] - base[:, 3]).max(), de.max(), de.mean()
    e1, e4, ee, em = drift(got)
    s1, s4, se, sm = drift(c32)
    assert abs(s1 - float(fx["stock_fp32_max_abs"][1])) < 1e-12 and abs(se - float(fx["stock_fp32_ema_max_after20"])) < 1e-12     # the yardstick IS the fixture's
    yard = {"stock float32 vs float64": (s1, s4, se, sm)}
    if "fp32_nodnn_curve" in fx.files:
        # the reference's second float32 realisation (oneDNN off).  The yardstick is the LARGEST distance between two of the reference's own runs:
        # each float32 run against the float64 run, and the two float32 runs against each other
        cn = fx["fp32_nodnn_curve"][:nsteps]
        yard["float32 (oneDNN off) vs float64"] = drift(cn)
        yard["stock float32 vs float32 (oneDNN off)"] = drift(c32, cn)
    y4, ye, ym = (max(y[i] for y in yard.values()) for i in (1, 2, 3))
    print(f"  {dt} vs the float64 reference over {nsteps} steps: loss1 max {e1:.2e}  loss4 max {e4:.2e}  EMA(total) after step 20: max {ee:.2e} mean {em:.2e}   "
          + "   ".join(f"[reference, {k}: {y[0]:.2e}  {y[1]:.2e}  {y[2]:.2e}  {y[3]:.2e}]" for k, y in yard.items()))
    e_got, e_ref = ema_of(got[:, 0]), ema_of(ref[:, 0])
    for s in (0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 250, 299):
        if s < nsteps:
            print(f"    step {s:3d}: total {got[s, 0]:+.4f} ({got[s, 0] - ref[s, 0]:+.1e}; stock fp32 {c32[s, 0] - ref[s, 0]:+.1e})  loss1 {got[s, 1]:.5f} ({got[s, 1] - ref[s, 1]:+.1e})  "
                  f"loss4 {got[s, 3]:.5f} ({got[s, 3] - ref[s, 3]:+.1e})  EMA {e_got[s]:+.4f} ({e_got[s] - e_ref[s]:+.1e})")
    assert e1 <= 1e-3, ("loss1 vs the float64 reference", e1)
    assert e4 <= LC_FACTOR * y4, ("loss4", e4, y4)
    assert ee <= LC_FACTOR * ye, ("EMA(total) after step 20, max", ee, ye)
    assert em <= LC_FACTOR * ym, ("EMA(total) after step 20, mean", em, ym)


# engine drift <= LC_FACTOR x the drift of the reference's own float32 runs (stock; oneDNN off), all measured against the reference's float64 run
# (no free-standing tolerance).  The chaotic quantities (EMA of the total) are ONE draw per run: see the measured ratios in
# profiles/r06_long_curve.txt before tightening.
LC_FACTOR = 2.0


def test_bf16_loss_curve_vs_rounding_aware_comparator(golden_dir):
    """north_star's "loss curve matching reference to 1e-3" for the BENCHMARKED dtype.  Against the float64 reference curve a bf16 run can only be
    held to 5e-3 .. 2.5e-2 on the total (test above): what separates them is what bf16 rounding does to the trajectory, not the kernels.  The
    comparator (oracle/pcrlv2_bf16_emulation.py: the pinned oracle's algorithm in float64 WITH the engine's rounding points; asserted equal to the
    oracle with rounding off) run for the same 12 steps on float32 master weights (oracle/make_emulated.py --curve ->
    tests/golden/e_curve_b8_32x32x16_12steps.npz) removes that: SURVEY App. C's sub-gates hold for bf16 --
      (i)   total loss on step 0 within 1e-3 (measured 3.3e-4).  Steps 1-2 stay at 5e-3 (measured 3.4e-3 / 2.0e-3): the remainder is NOT a rounding
            point the comparator lacks -- the MSE components agree to 1e-4 -- but the cosine terms, which reach the loss through BatchNorm1d over
            eight near-identical rows: float32-vs-float64 accumulation flips single bf16 roundings, and that path amplifies each flip (the
            float32 engine against the float64 reference shows the same from step 3 on);
      (iii) the MSE components over all 12 steps: restoration `loss1` within 3e-4 (measured 1.2e-4; against the float64 reference curve the gate
            is 1e-3), deep supervision `loss4` within 1e-3 on steps 0-5 and 3e-3 after (measured 1.1e-3 in round 4, 2.4e-3 at step 11 in round 5 -- see the
                note at the assertion; against the reference 4e-3)."""
    fx = np.load(os.path.join(golden_dir, "e_curve_b8_32x32x16_12steps.npz"))
    ref, b, dhw, nsteps = fx["curve"], int(fx["b"]), tuple(int(v) for v in fx["dhw"]), int(fx["nsteps"])
    batches = [O.fill_batch(b, dhw, dtype=torch.float32, seed=int(fx["batch_seed0"]) + s) for s in range(nsteps)]
    model = build(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=float(fx["base_lr"]), momentum=0.9, weight_decay=1e-4)
    random.seed(int(fx["seed"]))
    got = []
    for bt in batches:
        out = train_step(model, opt, bt, int(fx["epoch"]), MSELoss(), CosineSimilarityMean())
        got.append([float(v) for v in out])
    names = ("loss", "loss1", "loss2", "loss4", "local_loss")
    for s in range(nsteps):
        print(f"  bf16 vs comparator, step {s:2d}: " + "  ".join(f"{n} {got[s][i]:+.5f} ({got[s][i] - ref[s][i]:+.1e})" for i, n in enumerate(names)))
    # measured on MI355X (round 4): total 3.3e-4 / 3.4e-3 / 2.0e-3 on steps 0 / 1 / 2; loss1 <= 1.2e-4 and loss4 <= 1.1e-3 over all 12 steps
    assert abs(got[0][0] - ref[0][0]) < 1e-3, (0, "loss", got[0][0], ref[0][0])
    for s in (1, 2):
        assert abs(got[s][0] - ref[s][0]) < 5e-3, (s, "loss", got[s][0], ref[s][0])
    for s in range(nsteps):
        assert abs(got[s][1] - ref[s][1]) < 3e-4, (s, "loss1", got[s][1], ref[s][1])
        # (steps 6-11: 3e-3 since the data gradient takes the first BatchNorm-backward pass from its tiles -- a different summation ORDER of two
        #  per-channel sums; measured 2.4e-3 at step 11, 1.1e-3 before: the late steps of this curve follow the trajectory, not the kernels)
        assert abs(got[s][3] - ref[s][3]) < (1e-3 if s < 6 else 3e-3), (s, "loss4", got[s][3], ref[s][3])


def test_bf16_rounding_points_census(golden_dir):
    """A tripwire for the rounding-aware comparator (oracle/pcrlv2_bf16_emulation.py reproduces the engine's bf16 rounding points BY HAND, DESIGN
    section 2): every bf16 store of the engine happens inside a C-ABI launch that is called with dtype = bf16, so the census of those calls
    over one steady-state step (entry point -> number of calls) is frozen in tests/golden/bf16_call_census.json together with what each one
    rounds.  A new bf16-typed launch on the training path, or one called a different number of times, fails here: add the rounding point to the
    comparator (and to DESIGN section 2 and the fixture: tools/gates_probe.py --write-census) or show that it has none."""
    import json
    from pcrlv2_amd import _lib
    fx = json.load(open(os.path.join(golden_dir, "bf16_call_census.json")))
    L = _lib.lib()

    class Census:
        watch = {n for n, (_, args) in L.protos.items() if any(t == "pcrl_stream_t" for t, _ in args)}
        calls = {}

        def add(self, name, args):
            a = {an: v for (_, an), v in zip(L.protos[name][1], args)}
            if a.get("dtype") == _lib.PCRL_BF16:
                self.calls[name] = self.calls.get(name, 0) + 1
    model = build(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=7)
    random.seed(0)
    train_step(model, opt, batch, 3, MSELoss(), CosineSimilarityMean())
    torch.cuda.synchronize()
    c = Census()
    L.counter = c
    try:
        random.seed(0)
        train_step(model, opt, batch, 3, MSELoss(), CosineSimilarityMean())
        torch.cuda.synchronize()
    finally:
        L.counter = None
    got = dict(sorted(c.calls.items()))
    assert set(got) == set(fx["calls"]), ("bf16-typed launches changed", sorted(set(got) ^ set(fx["calls"])))
    assert got == fx["calls"], {k: (got[k], fx["calls"][k]) for k in got if got[k] != fx["calls"][k]}
    assert set(fx["rounding"]) == set(fx["calls"])        # every launch of the census says what it rounds


def test_config_c4_large_crops_step_properties():
    """BASELINE config C4: 128x128x64 crops, b=8, bf16 (LDS-halo / HBM stress: 8.4 M voxels per view, > 2^31 bytes per
    activation).  Properties: finite losses, every used parameter gets a finite gradient, 8 unused-head tensors get none,
    parameters move, and the step is bit-reproducible."""
    b = 8
    gen = torch.Generator().manual_seed(77)
    x1 = torch.randn(b, 1, 128, 128, 64, generator=gen)
    batch = (x1, x1 + 0.1 * torch.randn(b, 1, 128, 128, 64, generator=gen), torch.rand(b, 1, 128, 128, 64, generator=gen), None,
             [torch.randn(b, 1, 16, 16, 16, generator=gen) for _ in range(6)])
    res = []
    for rep in range(2):
        torch.manual_seed(0)
        model = PCRLv23d().to(DEV).train().set_compute_dtype(torch.bfloat16)
        opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        p0 = opt.flat_p.clone()
        random.seed(0)
        out = train_step(model, opt, batch, 0, MSELoss(), CosineSimilarityMean())
        vals = [float(v) for v in out]
        assert all(np.isfinite(vals)), vals
        assert sum(p.grad is None for p in model.parameters()) == 8
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        assert float((opt.flat_p - p0).abs().max()) > 0
        res.append((vals, opt.flat_p.clone()))
        del model, opt
        torch.cuda.empty_cache()
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_checkpoint_resume_continues_bit_exactly(dt, tmp_path):
    """a checkpoint of the reference's layout (train_3d.py:71-82: {'opt','state_dict','optimizer','epoch'} with
    torch.optim.SGD's state layout) written after two steps, loaded into a FRESH model + FusedSGD, must continue exactly like
    the uninterrupted run: parameters, BatchNorm buffers (running stats, num_batches_tracked) and momentum buffers after two
    more steps are bit-identical (the engine is deterministic; the python RNG state is carried by the test)."""
    from pcrlv2_amd.train_3d import load_checkpoint
    b, dhw = 2, (32, 32, 16)
    batches = [O.fill_batch(b, dhw, dtype=torch.float32, seed=40 + s) for s in range(4)]
    crit, cosine = MSELoss(), CosineSimilarityMean()

    def fresh():
        m = build(dt)
        return m, FusedSGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)

    random.seed(5)
    model, opt = fresh()
    for s in range(2):
        train_step(model, opt, batches[s], 3, crit, cosine)
    path = str(tmp_path / "pcrlv2_luna_pretask_1.0_3.pt")
    torch.save({"opt": None, "state_dict": model.state_dict(), "optimizer": opt.state_dict(), "epoch": 3}, path)
    rng = random.getstate()
    for s in range(2, 4):
        train_step(model, opt, batches[s], 3, crit, cosine)
    want_sd = {k: v.clone() for k, v in model.state_dict().items()}
    want_mom = opt.flat_buf.clone()

    model2, opt2 = fresh()
    train_step(model2, opt2, batches[3], 3, crit, cosine)      # dirty the fresh engine state first (packed weights, counters, momentum)
    assert load_checkpoint(path, model2, opt2) == 3
    random.setstate(rng)
    for s in range(2, 4):
        train_step(model2, opt2, batches[s], 3, crit, cosine)
    got_sd = model2.state_dict()
    assert list(got_sd) == list(want_sd)
    for k in want_sd:
        assert torch.equal(got_sd[k], want_sd[k]), k
    assert torch.equal(opt2.flat_buf, want_mom)
    # the optimizer state written by FusedSGD is torch.optim.SGD's: a stock SGD over the same parameters accepts it
    stock = torch.optim.SGD(model2.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    saved = torch.load(path, weights_only=False)["optimizer"]
    stock.load_state_dict(saved)
    assert len(stock.state) == len(saved["state"]) > 0   # (parameters that had no gradient yet have no buffer, as in torch)


DDP2_WORKER = r'''
import os, sys, random, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import pcrlv2_oracle as O
from pcrlv2_amd import ddp
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step
rank, world, _ = ddp.init_process_group_from_env("gloo")     # two processes, ONE GPU: gloo moves the CUDA buffers
torch.cuda.set_device(0)
dt = torch.bfloat16
batches = [O.fill_batch(2, (32, 32, 16), dtype=torch.float32, seed=70 + 10 * rank + s) for s in range(2)]   # different data per rank
finals = []
for overlap in (True, False):
    random.seed(3)                       # same scale draws on every rank
    model = PCRLv23d().cuda()
    model.load_state_dict(O.fill_state(torch.float32))
    model.train().set_compute_dtype(dt)
    opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    dp = ddp.DataParallel(model, opt, bucket_mb=8.0, overlap=overlap)
    assert dp._active and opt.grad_scale == 0.5 and len(dp.reducer.buckets) >= 3
    early = 0
    for bt in batches:
        losses = train_step(model, opt, bt, 3, MSELoss(), CosineSimilarityMean())
        assert all(torch.isfinite(l) for l in losses)
    finals.append(opt.flat_p.clone())
    assert not getattr(dp, "_warned_late", False), "a gradient arrived after its bucket was reduced"
# overlap on/off give the same parameters, and both ranks hold the same parameters
assert torch.equal(finals[0], finals[1]), (finals[0] - finals[1]).abs().max()
mine = finals[0].double().sum().reshape(1).cpu()
both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(both, mine)
assert both[0].item() == both[1].item(), both
# the step really used the other rank's data: a single-rank run on this rank's batches ends elsewhere
random.seed(3)
model = PCRLv23d().cuda(); model.load_state_dict(O.fill_state(torch.float32)); model.train().set_compute_dtype(dt)
opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
for bt in batches:
    train_step(model, opt, bt, 3, MSELoss(), CosineSimilarityMean())
assert not torch.equal(opt.flat_p, finals[0])
dist.barrier()
print("OK", rank, flush=True)
dist.destroy_process_group()      # tear the group down before the interpreter exits (gloo threads alive at exit abort the process now and then)
'''


def test_data_parallel_two_ranks_one_gpu_gloo(tmp_path):
    """The N > 1 path on real kernels: two processes (gloo, both on cuda:0) run the full model + FusedSGD + ddp.DataParallel on
    DIFFERENT batches.  Bucket overlap on/off give bit-identical parameters, both ranks end with the same parameters, no gradient
    arrives after its bucket was reduced, and the result differs from a single-rank run (the exchange happened)."""
    import subprocess
    import sys
    script = tmp_path / "ddp2.py"
    script.write_text(DDP2_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29761", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert all("OK" in o for o in outs)


DDP_ORACLE_WORKER = r"""
import os, sys, random, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import pcrlv2_oracle as O
from make_golden import sample_idx
from pcrlv2_amd import ddp
from pcrlv2_amd.models import PCRLv23d
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step
rank, world, _ = ddp.init_process_group_from_env("gloo")     # two processes, ONE GPU: gloo moves the CUDA buffers
torch.cuda.set_device(0)
fx = np.load(os.path.join(root, "tests", "golden", os.environ.get("TEST_DP_FIXTURE", "dp2_b4x2_32x32x16") + ".npz"))
b, dhw, nsteps = int(fx["meta/b_rank"]), tuple(int(v) for v in fx["meta/dhw"]), int(fx["meta/nsteps"])
assert world == int(fx["meta/world"])
assert ddp.chunk_partition_on() == (str(fx["meta/partition"]) == "chunk" if "meta/partition" in fx.files else False)
overlap = os.environ.get("TEST_OVERLAP", "0") == "1"
random.seed(int(fx["meta/seed"]))        # every rank draws the scales of the ONE global cos_loss call sequence
model = PCRLv23d().cuda()
model.load_state_dict(O.fill_state(torch.float32))
model.train().set_compute_dtype(torch.float32)
opt = FusedSGD(model.parameters(), lr=float(fx["meta/lr"]), momentum=0.9, weight_decay=1e-4)
dp = ddp.DataParallel(model, opt, bucket_mb=8.0, overlap=overlap)
assert dp._active and opt.grad_scale == 1.0 / world
for s in range(nsteps):
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=7 + 100 * s + 1000 * rank)       # make_golden.make_dp's rank batches
    out = train_step(model, opt, batch, int(fx["meta/epoch"]), MSELoss(), CosineSimilarityMean())
    for k, v in zip(("loss", "loss1", "loss2", "loss4", "local_loss"), out):
        tol = 2e-5 if (s == 0 or k in ("loss1", "loss4")) else 5e-4
        d = abs(float(v) - float(fx[f"step{s}/rank{rank}/{k}"]))
        assert d < tol, (rank, s, k, float(v), float(fx[f"step{s}/rank{rank}/{k}"]))
worst = 0.0
for name, p in model.named_parameters():
    f = p.detach().double().cpu().reshape(-1).numpy()
    d = np.abs(f[sample_idx(f.size, 64, 3)] - fx[f"final/{name}/samples"]).max()
    worst = max(worst, d)
    assert d < 5e-5, (rank, name, d)
if rank == 0:        # replica 0's running statistics are the ones that persist under nn.DataParallel
    model.flush_counters()
    sd = model.state_dict()
    for name in sd:
        if O.is_buffer(name):
            # after TWO updates: the second iteration's batch means are taken through parameters that already differ by up to 5e-5 and through
            # up to 16 batch-statistics normalisations (measured worst 1e-4 on down_tr128.ops.0.bn1.running_mean; after one forward set the
            # bound is 2e-6, test_fp32_step_matches_reference_golden).  Another rank's statistics, or statistics over the gathered batch,
            # differ from these by 1e-2 and more.
            np.testing.assert_allclose(sd[name].double().cpu().numpy(), fx[f"final_buf/{name}"], rtol=2e-3, atol=4e-4, err_msg=name)
dist.barrier()
print("OK", rank, "worst parameter |d| after %d data-parallel steps = %.2e" % (nsteps, worst), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_literal_chunk_partition_of_the_local_views(tmp_path, overlap):
    """PCRL_DP_LOCAL_PARTITION=chunk (opt-in; VERDICT r4 "missing" 5): nn.DataParallel's literal scatter of the concatenated [6B] local-view tensor
    (train_3d.py:121-123: with two replicas, replica 0 runs local views 0-2 of ALL samples, so the local pass's BatchNorm statistics group rows
    differently than the engine's by-sample default) -- two gloo ranks on one GPU against tests/golden/dp2chunk_b4x2_32x32x16.npz, which the REAL
    model produced under that scatter (oracle/make_golden.py --data-parallel): every rank's losses of both iterations, every parameter after the
    second update, rank 0's running statistics -- the same bounds as the by-sample test below.  The two fixtures differ (local_loss of step 0:
    -5.07e-3 / +3.07e-3 here, other values there), so passing this one with the default partition is impossible."""
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa, fb = (np.load(os.path.join(root, "tests", "golden", t + ".npz")) for t in ("dp2chunk_b4x2_32x32x16", "dp2_b4x2_32x32x16"))
    assert abs(float(fa["step0/rank0/local_loss"]) - float(fb["step0/rank0/local_loss"])) > 1e-4        # the partition changes the result
    script = tmp_path / "ddp_chunk.py"
    script.write_text(DDP_ORACLE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29775" if overlap else "29773", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               TEST_OVERLAP="1" if overlap else "0", TEST_DP_FIXTURE="dp2chunk_b4x2_32x32x16", PCRL_DP_LOCAL_PARTITION="chunk")
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert all("OK" in o for o in outs)
    print("\n".join(l for o in outs for l in o.splitlines() if l.startswith("OK")))


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_two_ranks_match_the_reference_dataparallel_oracle(tmp_path, overlap):
    """a14 PINNED: two ranks (gloo, both on cuda:0, float32, b = 4 per rank, two iterations) against tests/golden/dp2_b4x2_32x32x16.npz --
    the REAL reference model run with nn.DataParallel's semantics (train_3d.py:54,116-138: per-replica BatchNorm statistics, losses over
    the gathered batch, replicas' gradients summed, one SGD step; oracle/make_golden.py::make_dp).  Every rank's losses of both
    iterations, every parameter after the second update (5e-5, the single-process two-step tolerance) and rank 0's running statistics."""
    import subprocess
    import sys
    script = tmp_path / "ddp_oracle.py"
    script.write_text(DDP_ORACLE_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29771" if overlap else "29769", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               TEST_OVERLAP="1" if overlap else "0")
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert all("OK" in o for o in outs)
    print("\n".join(l for o in outs for l in o.splitlines() if l.startswith("OK")))


# ---- bf16 engine against the ROUNDING-AWARE comparator (oracle/pcrlv2_bf16_emulation.py) ----------------------------------------------
# The float64 golden differs from a bf16 step by what the roundings do to it; the comparator is the same float64 algorithm WITH the engine's
# roundings (weights, stored activations, stored gradients, composed weights), so what is left is float32-vs-float64 accumulation -- and any
# wrong tap, phase, border class or scale in a bf16-only kernel.  tests/golden/e_*.npz come from oracle/make_emulated.py (the comparator with
# rounding off is asserted to BE the pinned oracle there).  Measured on MI355X: see the tolerances' comments.
EMULATED = {
    # tag: (loss abs, map max abs, feature rel-L2, weight-tensor gradient norm worst, weight-tensor direction min,
    #       affine-vector gradient norm worst, affine-vector direction min, norm median over all tensors)
    # Measured on MI355X (e_b16): losses 3.8e-4 (the global cosine term; the MSE terms 2e-7 / 1e-6), maps 6e-3 .. 2.4e-2, features 2.1e-2 .. 2.7e-2;
    # gradient norms median 0.54 %; convolution / Linear weight tensors <= 3 % and direction >= 0.972; the per-channel BatchNorm vectors
    # (sums of dy * xhat over every voxel of the batch: cancellation) up to 8.5 % / 0.963; single-element tensors (the 1-channel heads'
    # BatchNorm parameters: one such sum) up to 25 %.  Against the float64 golden the same step is held to 25 % / 0.8 (GOLDEN_STEPS).
    # The direction floor of the weight tensors sits on bf16 rounding noise, not on the kernels: the SAME step with the 2^3-level gather
    # convolution walking its K steps in a different order (PCRL_IGEMM_VMAJOR=0 / 8: identical products, other float32 split points) moves the
    # lowest tensor between 0.9713 (down_tr128.ops.0) and 0.9666 (up_tr64.ops.1) while its norm error IMPROVES (2.38 -> 2.25 %) and the median
    # stays 0.65 %.  Gate 0.96: a wrong tap / phase / scale lands below 0.9 (the float64 golden's gate, GOLDEN_STEPS, is 0.8).
    "e_b16_32x32x16": (8e-4, 4e-2, 4e-2, 0.03, 0.96, 0.10, 0.95, 0.01),
    # BASELINE crop size, b = 8: measured losses 8.5e-5, maps 9e-3 .. 2.5e-2, features 1.3e-2 .. 2.5e-2; weight-tensor norms <= 0.94 % (!),
    # BatchNorm vectors <= 7.1 %, median 0.41 %; DIRECTIONS 0.965 (weights) / 0.911 (up_tr64.ops.0.bn1.bias): the cosine terms reach the
    # decoder through BatchNorm1d over EIGHT rows of near-identical global averages, which turns the 2e-4 relative differences that rare
    # one-ulp rounding flips leave in the activations into percent-level differences of that part of the gradient (the same mechanism makes
    # the features differ by 2e-2 here and by 6e-2 from the float64 golden); the norm gates are unaffected.
    "e_luna_b8_64x64x32": (8e-4, 4e-2, 4e-2, 0.03, 0.95, 0.10, 0.90, 0.01),
}
WEIGHT_TENSORS = ("conv1.weight", "up_conv.weight", "predictor_head.0.weight", "predictor_head.3.weight", "final_conv.weight")


@pytest.mark.parametrize("tag", list(EMULATED))
def test_bf16_step_against_the_rounding_aware_comparator(tag, golden_dir):
    path = os.path.join(golden_dir, tag + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{tag}.npz not generated (oracle/make_emulated.py)")
    fx = np.load(path)
    tol_loss, tol_map, tol_feat, w_norm, w_dir, a_norm, a_dir, tol_med = EMULATED[tag]
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=int(fx["meta/batch_seed"]))
    model = build(torch.bfloat16)
    r = forward_losses(model, batch, int(fx["meta/epoch"]), int(fx["meta/seed"]))
    rep = {}
    for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
        rep[k] = abs(float(r[k].detach()) - float(fx[f"step0/{k}"]))
    rep["out"] = float(np.abs(samples(r["mask1"], 1024) - fx["fwd/out/samples"]).max())
    for i in range(3):
        rep[f"mid{i}"] = float(np.abs(samples(r["mid1"][i], 1024) - fx[f"fwd/mid{i}/samples"]).max())
        for j, nm in enumerate(("pro", "pre")):
            a, ref = r["dec1"][i][j].detach().double().cpu(), torch.from_numpy(fx[f"fwd/{nm}{i}"]).double()
            rep[f"{nm}{i}"] = ((a - ref).norm() / ref.norm()).item()
    r["loss"].backward()
    rows = []      # (norm deviation, direction or None, name, class)
    for name, p in model.named_parameters():
        if f"grad/{name}/none" in fx.files:
            assert p.grad is None, name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        if name.endswith(ZERO_GRAD):
            continue
        l2 = float(fx[f"grad/{name}/l2"])
        dev = abs(float(p.grad.double().norm()) - l2) / l2
        cls = "weight" if name.endswith(WEIGHT_TENSORS) and p.numel() >= 16 else ("vector" if p.numel() >= 16 else "single")
        d = None
        if p.numel() >= 16:
            g, ref = samples(p.grad, 2048), fx[f"grad/{name}/samples"]
            d = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-300))
        rows.append((dev, d, name, cls))
    med = sorted(x[0] for x in rows)[len(rows) // 2]
    print(f"\n[{tag}] " + ", ".join(f"{k}={v:.2e}" for k, v in rep.items()))
    for cls in ("weight", "vector", "single"):
        sel = [x for x in rows if x[3] == cls]
        worst = sorted(sel, key=lambda x: -x[0])[:3]
        lows = sorted((x for x in sel if x[1] is not None), key=lambda x: x[1])[:3]
        print(f"[{tag}] {cls:6s} ({len(sel)} tensors): norm worst {[(round(x[0], 4), x[2]) for x in worst]}; direction lowest {[(round(x[1], 4), x[2]) for x in lows]}")
    print(f"[{tag}] gradient-norm median over all tensors {med:.4f}")
    assert all(rep[k] <= tol_loss for k in ("loss", "loss1", "loss2", "loss4", "local_loss")), rep
    assert rep["out"] <= tol_map and all(rep[f"mid{i}"] <= tol_map for i in range(3)), rep
    assert all(rep[f"{nm}{i}"] <= tol_feat for i in range(3) for nm in ("pro", "pre")), rep
    assert med <= tol_med, med
    for dev, d, name, cls in rows:
        if cls == "weight":
            assert dev <= w_norm and d >= w_dir, (name, dev, d)
        elif cls == "vector":
            assert dev <= a_norm and d >= a_dir, (name, dev, d)
        else:       # one element: a single cancellation-prone sum
            assert dev <= 0.35, (name, dev)


def test_unused_outputs_of_view_2_and_local_passes_are_skipped_without_a_trace():
    """train_3d.SKIP_UNUSED_OUTPUTS (default on): the second view's and the local views' reconstruction and upsampled deep-supervision maps
    are never used by the step (train_3d.py:117,123) and have no state -- the engine's model does not compute them
    (forward(..., features_only=True)).  Parameters, momentum, running statistics AND num_batches_tracked after two steps must be
    bit-identical to computing them, and features_only must return the same features as the full forward."""
    from pcrlv2_amd import train_3d
    batches = [O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=31 + s) for s in range(2)]
    finals = []
    keep = train_3d.SKIP_UNUSED_OUTPUTS
    try:
        for on in (True, False):
            train_3d.SKIP_UNUSED_OUTPUTS = on
            model = build(torch.bfloat16)
            opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
            random.seed(5)
            for bt in batches:
                losses = train_step(model, opt, bt, 3, MSELoss(), CosineSimilarityMean())
            torch.cuda.synchronize()
            finals.append(([float(l) for l in losses], opt.flat_p.clone(), opt.flat_buf.clone(), {k: v.clone() for k, v in model.state_dict().items() if O.is_buffer(k)}))
    finally:
        train_3d.SKIP_UNUSED_OUTPUTS = keep
    (la, pa, ma, ba), (lb, pb, mb, bb) = finals
    assert la == lb and torch.equal(pa, pb) and torch.equal(ma, mb)
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k
    model = build(torch.float32)
    x = batches[0][0].to(DEV)
    with torch.no_grad():
        out, feats, masks = model(x)
        model.load_state_dict(O.fill_state(torch.float32))
        out2, feats2, masks2 = model(x, features_only=True)
    assert out2 is None and masks2 == [] and out is not None and len(masks) == 3
    for (a, b), (c, d) in zip(feats, feats2):
        assert torch.equal(a, c) and torch.equal(b, d)


def test_no_device_malloc_after_the_first_step_at_the_bench_size():
    """ops.provision_allocator: the caching allocator's per-stream pools are sized once, after the first complete step of a batch shape, so
    that the multi-stream steady state (blocks in flight across the two-step run-ahead window) needs no hipMalloc later -- a short
    benchmark run (`--steps 20 --warmup 5`) used to hold 21-23 device mallocs inside its timed region."""
    from bench import synthetic_batch
    dev = torch.device(DEV)
    # start from the allocator state of a fresh process (what bench.py sees): the pools and the provisioning record of the tests that ran
    # before this one in the same process are dropped
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ops._provisioned.clear()
    ops._provisioned_segments.clear()
    batch = synthetic_batch(32, (64, 64, 32), 16, dev, 11)
    torch.manual_seed(0)
    random.seed(0)
    model = PCRLv23d().to(DEV).train().set_compute_dtype(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    crit, cosine = MSELoss(), CosineSimilarityMean()
    for _ in range(2):
        train_step(model, opt, batch, 0, crit, cosine, guard=False)
    torch.cuda.synchronize()
    n0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    for _ in range(15):
        train_step(model, opt, batch, 0, crit, cosine, guard=False)
    torch.cuda.synchronize()
    grown = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n0
    # without the provisioning: 21-23 in a run of this length; a step whose draws need a block size the first step never used may still add one
    assert grown <= 3, f"{grown} device mallocs in 15 steps after the pools were provisioned"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_lazy_skip_attributes_are_the_stored_tensors_and_leave_the_step_bit_identical(dt):
    """The reference stashes the encoder stages' unpooled outputs as `self.skip_out64 / 128 / 256` (pcrlv2_model_3d.py:114-117) and never reads them
    (UpTransition takes no skip input).  The engine's training step does not store them (forward(..., lazy_skips=True): the fused normalise + pool
    pass writes the pooled tensor only -- 47 % of that pass's HBM bytes); the attribute still answers, built from the stage's saved
    pre-normalisation tensor when read.  Held: (i) every lazily built attribute is bit-equal to the tensor the plain forward stores, same shape,
    dtype and NDHWC layout, and `out512` (a stage output that IS consumed) is unchanged; (ii) outputs, features and maps of the two forwards are
    bit-equal; (iii) two training steps give bit-identical losses, parameters, momentum and BatchNorm buffers whether train_3d passes
    lazy_skips or not."""
    from pcrlv2_amd import train_3d
    from pcrlv2_amd.models import pcrlv2_model_3d as M3
    batch = O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=77)
    x = batch[0].to(DEV)
    model = build(dt)
    with torch.no_grad():
        out_a, feats_a, masks_a = model(x)
        stored = {n: getattr(model, n).clone() for n in ("skip_out64", "skip_out128", "skip_out256", "out512")}
        model.load_state_dict(O.fill_state(torch.float32))
        out_b, feats_b, masks_b = model(x, lazy_skips=True)
        assert all(isinstance(model.__dict__["_skip_store"][n], M3._LazySkip) for n in ("skip_out64", "skip_out128", "skip_out256"))
        for n, want in stored.items():
            got = getattr(model, n)
            assert torch.is_tensor(got) and got.shape == want.shape and got.dtype == want.dtype and got.stride() == want.stride(), n
            assert torch.equal(got, want), n
            assert getattr(model, n) is got             # materialised once
    assert torch.equal(out_a, out_b) and all(torch.equal(a, b) for a, b in zip(masks_a, masks_b))
    for (a, b), (c, d) in zip(feats_a, feats_b):
        assert torch.equal(a, c) and torch.equal(b, d)
    batches = [O.fill_batch(4, (32, 32, 16), dtype=torch.float32, seed=41 + s) for s in range(2)]
    finals = []
    keep = train_3d.LAZY_SKIPS
    for lazy in (True, False):
        train_3d.LAZY_SKIPS = lazy
        try:
            m = build(dt)
            opt = FusedSGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
            random.seed(9)
            for bt in batches:
                losses = train_step(m, opt, bt, 3, MSELoss(), CosineSimilarityMean())
            torch.cuda.synchronize()
            finals.append(([float(v) for v in losses], opt.flat_p.clone(), opt.flat_buf.clone(), {k: v.clone() for k, v in m.state_dict().items() if O.is_buffer(k)}))
        finally:
            train_3d.LAZY_SKIPS = keep
    (la, pa, ma, ba), (lb, pb, mb, bb) = finals
    assert la == lb and torch.equal(pa, pb) and torch.equal(ma, mb)
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k


def test_per_epoch_empty_cache_keeps_every_provisioned_segment():
    """The reference returns the caching allocator's memory to the driver after every epoch (train_3d.py:83); ops.empty_cache does the same but holds
    the per-stream pools the step's steady state needs (ops.provision_allocator) across the call -- re-reserving them stalls the device for seconds.
    ADVICE r5 #5 / round 6: a placeholder sized from a provisioned segment's hole could land in another segment (equal-size holes; blocks still
    "active_pending_free" at snapshot time turning free under the placeholders) and one provisioned segment was released per call.  Held here: after
    warm-up steps, three rounds of (steps, empty_cache): every provisioned segment is still reserved, memory that is NOT provisioned does go back
    (reserved bytes do not grow from call to call), and the steps after a call make no device malloc."""
    dev = torch.device(DEV)
    b, dhw = 8, (32, 32, 16)
    model = build(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=3)
    crit, cos = MSELoss(), CosineSimilarityMean()
    random.seed(0)
    for _ in range(6):
        train_step(model, opt, batch, 0, crit, cos)
    torch.cuda.synchronize()
    idx = torch.cuda.current_device()
    mine = {e[1] for e in ops._provisioned_segments if e[0] == idx}
    assert mine, "the allocator pools were not provisioned after the first steps"
    reserved = []
    for rnd in range(3):
        scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)      # something cached outside the pools, for empty_cache to give back
        del scratch
        ops.empty_cache(dev)
        left = {s["address"] for s in torch.cuda.memory_snapshot() if s["device"] == idx}
        assert mine <= left, (rnd, len(mine - left), "provisioned segments released by ops.empty_cache")
        reserved.append(torch.cuda.memory_reserved(dev))
        n0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        for _ in range(3):
            train_step(model, opt, batch, 0, crit, cos)
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats(dev).get("num_device_alloc", 0) == n0, (rnd, "device mallocs in the steps after empty_cache")
    assert reserved[2] <= reserved[0], reserved
