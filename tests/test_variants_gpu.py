"""Constructor variants of PCRLv23d that the reference accepts but train_3d.py:45 never instantiates (models/pcrlv2_model_3d.py:15-16,
22-25,98): act in {'elu', 'prelu'}, norm='in', in_channels != 1, n_class != 1 -- the HIP engine against tests/golden/v_*.npz, which
oracle/make_golden.py --variants wrote from the REAL reference built with the same arguments (float64, oneDNN off): one train-mode
forward and the gradients of O.variant_loss (a fixed closed-form functional of every output) w.r.t. every parameter.

Tolerances (float32 mode; same envelope as tests/test_model_gpu.py): sigmoid maps 5e-5 abs, [b, C] features 3e-4 abs (InstanceNorm over the
2 x 2 x 1 voxels of the deepest level and BatchNorm1d over three rows amplify rounding), the scalar 2e-5, gradients per-tensor rel-L2 2e-2
(analytically zero ones 1e-5 abs), running statistics 1e-5.  bfloat16 mode: the forward only (maps 5e-2, feature direction 0.98)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pcrlv2_oracle as O  # noqa: E402
from make_golden import VARIANTS, sample_idx  # noqa: E402
from pcrlv2_amd.models import PCRLv23d  # noqa: E402

DEV = "cuda"


def samples(t, k, seed=3):
    f = t.detach().double().cpu().reshape(-1).numpy()
    return f[sample_idx(f.size, k, seed)]


def okw(kw):
    return dict(n_class=kw.get("n_class", 1), in_channels=kw.get("in_channels", 1), act=kw.get("act", "relu"), norm=kw.get("norm", "bn"))


def run(tag, dtype, golden_dir):
    kw = VARIANTS[tag]
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    k = okw(kw)
    st = O.fill_state(torch.float32, **k)
    model = PCRLv23d(**kw).to(DEV)
    assert list(model.state_dict().keys()) == list(st.keys()), "state_dict differs from the reference's layout"
    model.load_state_dict(st)
    model.train()
    model.set_compute_dtype(dtype)
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    x = O.variant_input(b, dhw, k["in_channels"], torch.float32).to(DEV)
    out, feats, masks = model(x)
    return fx, model, out, feats, masks


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_variant_fp32_matches_reference_golden(tag, golden_dir):
    fx, model, out, feats, masks = run(tag, torch.float32, golden_dir)
    assert tuple(out.shape)[:2] == (int(fx["meta/b"]), int(fx["meta/n_class"]))
    np.testing.assert_allclose(samples(out, 512), fx["fwd/out/samples"], rtol=0, atol=5e-5)
    for i in range(3):
        np.testing.assert_allclose(feats[i][0].detach().cpu().numpy(), fx[f"fwd/pro{i}"], rtol=0, atol=3e-4)
        np.testing.assert_allclose(feats[i][1].detach().cpu().numpy(), fx[f"fwd/pre{i}"], rtol=0, atol=3e-4)
        np.testing.assert_allclose(samples(masks[i], 256), fx[f"fwd/mask{i}/samples"], rtol=0, atol=5e-5)
    L = O.variant_loss(out, feats, masks)
    assert abs(float(L.detach()) - float(fx["loss"])) < 2e-5
    L.backward()
    torch.cuda.synchronize()
    bad = []
    for name, p in model.named_parameters():
        if f"grad/{name}/none" in fx.files:
            assert p.grad is None, name
            continue
        assert p.grad is not None, f"{name}: missing gradient"
        ref_s, l2 = fx[f"grad/{name}/samples"], float(fx[f"grad/{name}/l2"])
        got_s, got_l2 = samples(p.grad, 64), float(p.grad.double().norm())
        if l2 < 1e-9:       # a bias in front of a batch- / instance-statistics normalisation: analytically zero
            assert np.abs(got_s).max() <= 1e-5, (name, np.abs(got_s).max())
            continue
        rel = np.linalg.norm(got_s - ref_s) / max(np.linalg.norm(ref_s), 1e-30)
        if rel > 2e-2 or abs(got_l2 - l2) / l2 > 2e-2:
            bad.append((name, float(rel), abs(got_l2 - l2) / l2))
    assert not bad, f"gradients outside tolerance: {bad[:8]} (+{max(0, len(bad) - 8)} more)"
    sd = model.state_dict()
    for key in fx.files:
        if key.startswith("buf1/"):
            np.testing.assert_allclose(sd[key[5:]].double().cpu().numpy(), fx[key], rtol=0, atol=1e-5, err_msg=key)


@pytest.mark.parametrize("tag", ["v_elu", "v_all"])
def test_variant_bf16_forward(tag, golden_dir):
    fx, model, out, feats, masks = run(tag, torch.bfloat16, golden_dir)
    assert np.abs(samples(out, 512) - fx["fwd/out/samples"]).max() < 5e-2
    for i in range(3):
        for j, nm in enumerate(("pro", "pre")):
            a, r = feats[i][j].detach().double().cpu().numpy().ravel(), fx[f"fwd/{nm}{i}"].ravel()
            assert a @ r / (np.linalg.norm(a) * np.linalg.norm(r)) > 0.98, (nm, i)
    O.variant_loss(out, feats, masks).backward()     # the backward runs (values are pinned in float32 mode)
    torch.cuda.synchronize()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_variant_eval_mode_runs_and_matches_oracle(golden_dir):
    """model.eval() of a variant (running statistics where the variant has them) against the float64 oracle in eval mode."""
    kw = VARIANTS["v_all"]
    k = okw(kw)
    st = O.fill_state(torch.float32, **k)
    model = PCRLv23d(**kw).to(DEV)
    model.load_state_dict(st)
    model.eval()
    x = O.variant_input(2, (32, 32, 16), k["in_channels"], torch.float32)
    out, feats, masks = model(x.to(DEV))
    st64 = {n: (v.double() if v.is_floating_point() else v) for n, v in st.items()}
    with torch.no_grad():
        r_out, r_feats, r_masks = O.forward(st64, x.double(), training=False, act=k["act"], norm=k["norm"])
    assert (out.double().cpu() - r_out).abs().max() < 5e-5
    for i in range(3):
        assert (feats[i][0].double().cpu() - r_feats[i][0]).abs().max() < 3e-4
        assert (masks[i].double().cpu() - r_masks[i]).abs().max() < 5e-5


@pytest.mark.parametrize("how", ["torch_sgd", "manual_copy", "eval_between"])
def test_padded_first_layer_follows_weight_updates_that_bypass_the_engine_optimizer(how):
    """in_channels != 1 zero-pads the first convolution's weight; the packed-weight cache must follow the REAL parameter however
    it is updated -- torch.optim.SGD's in-place update, a manual copy_ between two forwards (train or eval) -- not only FusedSGD's epoch bump.
    Two forwards around an update of down_tr64.ops.0.conv1.weight, each against the float64 oracle on the state of that moment."""
    kw = dict(in_channels=3)
    k = okw(kw)
    st = O.fill_state(torch.float32, **k)
    if how == "eval_between":
        # eval mode normalises with the RUNNING statistics: with the initial ones (variance 1 against activations of ~0.05) every layer shrinks its
        # input and the outputs barely depend on the first layer (the oracle's own output moves by 1e-5 under the update below) -- statistics of
        # the activations' real size make the comparison sensitive (asserted below)
        for key in st:
            if key.endswith("running_var"):
                st[key] = torch.full_like(st[key], 0.25)
    model = PCRLv23d(**kw).to(DEV)
    model.load_state_dict(st)
    model.set_compute_dtype(torch.float32)
    x = O.variant_input(3, (32, 32, 16), 3, torch.float32)
    name = "down_tr64.ops.0.conv1.weight"

    def engine(train, keep=None):
        model.train(train)
        out, feats, _masks = model(x.to(DEV))
        if keep is not None:
            keep.append(out)
        return [out.detach().double().cpu()] + [f[0].detach().double().cpu() for f in feats]

    def snapshot():
        return {n: (v.detach().double().cpu() if v.is_floating_point() else v.detach().cpu()) for n, v in model.state_dict().items()}

    def oracle(sd, train):
        with torch.no_grad():
            out, feats, _ = O.forward(sd, x.double(), training=train, act=k["act"], norm=k["norm"])
        return [out] + [f[0] for f in feats]

    def close(got, ref, what):       # absolute for O(1) tensors, relative to the tensor's size where the running statistics above blow the features up
        for i, (a, b) in enumerate(zip(got, ref)):
            assert (a - b).abs().max() < (5e-5 if i == 0 else 3e-4) * max(1.0, float(b.abs().max())), (what, i, float((a - b).abs().max()), float(b.abs().max()))

    train = how != "eval_between"
    model.train(train)
    sd_before = snapshot()
    ref0 = oracle(sd_before, train)
    kept = []
    got0 = engine(train, kept)
    close(got0, ref0, "before the update")
    w = dict(model.named_parameters())[name]
    if how == "torch_sgd":
        opt = torch.optim.SGD(model.parameters(), lr=0.5)
        (kept[0] * kept[0]).mean().backward()
        torch.cuda.synchronize()
        assert w.grad is not None and float(w.grad.abs().max()) > 0
        opt.step()
    else:
        with torch.no_grad():
            w.copy_(w * -2.0 + 0.05)
    model.flush_counters()
    sd_mid = snapshot()
    assert (sd_mid[name] - sd_before[name]).abs().max() > 1e-4      # the parameter really moved
    ref1 = oracle(sd_mid, train)
    # the comparison below can only catch a stale pack if the update is visible in what is compared: the oracle's own outputs must have moved
    moved = max(float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(ref1, ref0))
    assert moved > 1e-3, moved
    got1 = engine(train)
    close(got1, ref1, "after the update: the first layer kept the packed weights of the previous forward?")
