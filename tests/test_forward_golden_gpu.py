"""The step's three forwards at the EXACT BASELINE batches against the real reference.

tests/golden/f_c2_b32_64x64x32.npz and f_c4_b8_128x128x64.npz hold what the REAL reference model (models/pcrlv2_model_3d.py in train mode,
float64, oneDNN off) and the imported `cos_loss` produce for BASELINE config C2 (b = 32, 64x64x32) and C4 (b = 8, 128x128x64) under
torch.no_grad() -- float64 backward does not fit the authoring container at these sizes, the forwards do (oracle/make_golden.py
`make_forward`): samples of the reconstruction and of the three deep-supervision maps, the six [b, C] feature tensors of view 1, all five
loss terms and the BatchNorm running statistics after the three passes.

Stated tolerances (absolute unless noted; float32 ones are SURVEY App. C's forward envelope of stock PyTorch float32 against float64):
  float32 : sigmoid maps 5e-5, features 2e-4, losses 1e-5, running statistics 1e-5 relative to the largest entry of the tensor.
  bfloat16: activations carry 8 mantissa bits through 17 convolution + BatchNorm layers: maps 1e-1 max / 2e-2 mean, features by direction
            (cosine >= 0.995 per scale), losses 3e-3, running statistics 1.5e-2 relative to the largest entry.
Measured on MI355X (C2 / C4): float32 losses 6e-7 / 6e-7, maps 1.2e-5 / 1.9e-5, features 5.6e-5 / 4.8e-5, running statistics 8e-7 / 1.2e-6;
bfloat16 losses 1.1e-3 / 1.1e-3 (the global cosine term; MSE terms 3e-6), maps max 6.3e-2 / 7.6e-2 (the full-resolution deep-supervision map)
and mean 1.1e-2 / 1.4e-2, feature cosine >= 0.998 / 0.996 (relative L2 5-9e-2), running statistics 3.7e-3 / 5.7e-3."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pcrlv2_oracle as O  # noqa: E402
from make_golden import sample_idx  # noqa: E402
from pcrlv2_amd.models import PCRLv23d  # noqa: E402
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, begin_step, step_losses  # noqa: E402

DEV = "cuda"
FIXTURES = ["f_c2_b32_64x64x32", "f_c4_b8_128x128x64"]
LOSSES = ("loss", "loss1", "loss2", "loss4", "local_loss")


def _samples(t, k):
    f = t.detach().double().cpu().reshape(-1).numpy()
    return f[sample_idx(f.size, k, 3)]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", FIXTURES)
def test_three_forwards_at_the_baseline_batch_match_the_reference(tag, dt, golden_dir):
    path = os.path.join(golden_dir, tag + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{tag}.npz not generated yet (oracle/make_golden.py --forward-only)")
    fx = np.load(path)
    b, dhw, epoch, seed = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"]), int(fx["meta/epoch"]), int(fx["meta/seed"])
    batch = O.fill_batch(b, dhw, dtype=torch.float32, seed=int(fx["meta/batch_seed"]))
    model = PCRLv23d().to(DEV)
    model.load_state_dict(O.fill_state(torch.float32))
    model.train().set_compute_dtype(dt)
    bf = dt == torch.bfloat16

    # the product's own step function (three forwards on three streams, all cosine terms in one launch), no backward
    random.seed(seed)
    begin_step()
    with torch.no_grad():
        got = step_losses(model, batch, epoch, MSELoss(), CosineSimilarityMean())
    torch.cuda.synchronize()
    got = dict(zip(LOSSES, (float(v) for v in got)))
    rep = {}
    for k in LOSSES:
        rep[k] = abs(got[k] - float(fx["step0/" + k]))
    tol_loss = 3e-3 if bf else 1e-5
    assert all(rep[k] <= tol_loss for k in LOSSES), (rep, got)

    # the same forwards again through the public model API (eleven tensors out), from the same state, for the tensors themselves
    model.load_state_dict(O.fill_state(torch.float32))
    random.seed(seed)
    begin_step()
    x1, x2, _gt, _, loc = batch
    with torch.no_grad():
        out, feats, mids = model(x1.to(DEV))
        model(x2.to(DEV))
        model(torch.cat([v.to(DEV) for v in loc], 0), local=True)
    torch.cuda.synchronize()
    d_out = np.abs(_samples(out, 1024) - fx["fwd/out/samples"])
    rep["out max"], rep["out mean"] = d_out.max(), d_out.mean()
    rep["out l2 rel"] = abs(float(out.double().norm()) - float(fx["fwd/out/l2"])) / float(fx["fwd/out/l2"])
    for i in range(3):
        d = np.abs(_samples(mids[i], 1024) - fx[f"fwd/mid{i}/samples"])
        rep[f"mid{i} max"], rep[f"mid{i} mean"] = d.max(), d.mean()
        for j, nm in enumerate(("pro", "pre")):
            ref = torch.from_numpy(fx[f"fwd/{nm}{i}"]).double()
            g = feats[i][j].double().cpu()
            rep[f"{nm}{i} max"] = (g - ref).abs().max().item()
            rep[f"{nm}{i} cos"] = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item()
            rep[f"{nm}{i} relL2"] = ((g - ref).norm() / ref.norm()).item()
    sd = model.state_dict()
    worst_stat = 0.0
    for k in fx.files:
        if not k.startswith("buf1/"):
            continue
        name = k[5:]
        ref = torch.from_numpy(np.asarray(fx[k])).double()
        g = sd[name].double().cpu()
        if name.endswith("num_batches_tracked"):
            assert int(g) == int(ref), name
            continue
        worst_stat = max(worst_stat, ((g - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item())
    rep["running stats rel"] = worst_stat
    print(f"\n[{tag} {'bf16' if bf else 'fp32'}] " + ", ".join(f"{k}={v:.3g}" for k, v in rep.items()))

    if not bf:
        assert rep["out max"] <= 5e-5 and all(rep[f"mid{i} max"] <= 5e-5 for i in range(3)), rep
        assert all(rep[f"{nm}{i} max"] <= 2e-4 for i in range(3) for nm in ("pro", "pre")), rep
        assert rep["running stats rel"] <= 1e-5, rep
    else:
        assert rep["out max"] <= 1e-1 and rep["out mean"] <= 2e-2, rep
        assert all(rep[f"mid{i} max"] <= 1e-1 and rep[f"mid{i} mean"] <= 2e-2 for i in range(3)), rep
        assert all(rep[f"{nm}{i} cos"] >= 0.995 for i in range(3) for nm in ("pro", "pre")), rep
        assert rep["running stats rel"] <= 1.5e-2, rep
