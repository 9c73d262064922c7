"""CPU: the oracle (oracle/pcrlv2_oracle.py) against the golden vectors generated from the REAL reference
(oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP path with both."""
import os
import random

import numpy as np
import pytest
import torch

import pcrlv2_oracle as O
from make_golden import sample_idx


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, tag + ".npz"))


def _samples(t, k, seed=3):
    f = t.detach().double().reshape(-1).numpy()
    return f[sample_idx(f.size, k, seed)]


@pytest.fixture(scope="module")
def small(golden_dir):
    fx = _load(golden_dir, "c_small_b4_32x32x16")
    b, dhw, nsteps = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"]), int(fx["meta/nsteps"])
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    st0 = O.fill_state(torch.float64)
    batches = [O.fill_batch(b, dhw, dtype=torch.float64, seed=7 + 100 * s) for s in range(nsteps)]
    with torch.backends.mkldnn.flags(enabled=False):
        st, mom, log, g0 = O.train_steps(st0, batches, int(fx["meta/epoch"]), float(fx["meta/base_lr"]), int(fx["meta/epochs"]), int(fx["meta/seed"]))
        rng = random.Random(int(fx["meta/seed"]))
        nb = {}
        fwd = O.step_losses(st0, batches[0], int(fx["meta/epoch"]), rng, nb)
    return fx, st, log, g0, fwd, nb


def test_layout_matches_reference_manifest(golden_dir):
    lines = [l.split(" ", 1) for l in open(os.path.join(golden_dir, "state_dict_manifest.txt")).read().strip().splitlines()]
    lay = O.state_layout()
    assert [k for k, _ in lines] == list(lay.keys())
    for (k, rest), shape in zip(lines, lay.values()):
        assert rest.startswith(str(tuple(shape))), (k, rest, shape)
    n_params = sum(int(np.prod(s)) if s else 1 for k, s in lay.items() if not O.is_buffer(k))
    assert n_params == 17111434


def test_losses_match_reference(small):
    fx, st, log, g0, fwd, nb = small
    for s, l in enumerate(log):
        for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
            assert abs(l[k] - float(fx[f"step{s}/{k}"])) < 1e-10, (s, k)
        assert l["index2"] == int(fx[f"step{s}/index2"])


def test_forward_matches_reference(small):
    fx, st, log, g0, fwd, nb = small
    np.testing.assert_allclose(_samples(fwd["mask1"], 256), fx["fwd/out/samples"], rtol=0, atol=1e-10)
    for i in range(3):
        np.testing.assert_allclose(fwd["dec1"][i][0].detach().numpy(), fx[f"fwd/pro{i}"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(fwd["dec1"][i][1].detach().numpy(), fx[f"fwd/pre{i}"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(_samples(fwd["mid1"][i], 256), fx[f"fwd/mid{i}/samples"], rtol=0, atol=1e-10)


def test_gradients_match_reference(small):
    fx, st, log, g0, fwd, nb = small
    n_none = 0
    for name, g in g0.items():
        if g is None:
            assert f"grad/{name}/none" in fx.files
            n_none += 1
            continue
        l2 = float(fx[f"grad/{name}/l2"])
        np.testing.assert_allclose(_samples(g, 64), fx[f"grad/{name}/samples"], rtol=1e-8, atol=1e-9 * max(l2, 1e-3))
    assert n_none == 8  # the two deep-supervision heads not selected by index2 (4 tensors each), SURVEY Q3


def test_state_after_two_steps_matches_reference(small):
    fx, st, log, g0, fwd, nb = small
    for name, v in st.items():
        if O.is_buffer(name):
            continue
        np.testing.assert_allclose(_samples(v, 64), fx[f"final/{name}/samples"], rtol=1e-9, atol=1e-11)
    for name, v in nb.items():
        np.testing.assert_allclose(v.double().numpy(), fx[f"buf1/{name}"], rtol=1e-9, atol=1e-11)


def test_lr_schedule(golden_dir):
    lrs = np.load(os.path.join(golden_dir, "lr_schedule.npz"))["lr"]
    mine = np.array([O.lr_at(e, 1e-3, 240) for e in range(241)])
    np.testing.assert_allclose(mine, lrs, rtol=0, atol=1e-18)
    assert mine[-1] < 1e-18 and mine[0] == 1e-3


def test_cosine_similarity_semantics():
    x = torch.tensor([[3.0, 4.0], [0.0, 0.0]], dtype=torch.float64)
    y = torch.tensor([[4.0, 3.0], [1.0, 0.0]], dtype=torch.float64)
    ref = torch.nn.CosineSimilarity()(x, y)
    np.testing.assert_allclose(O.cosine_similarity(x, y).numpy(), ref.numpy(), atol=1e-15)


def test_eval_mode_forward_matches_reference(golden_dir):
    """oracle.forward(training=False) against the REAL reference in .eval() (tests/golden/eval_b2_32x32x16.npz): running statistics
    moved by one training step, then an inference forward -- what a consumer of the checkpoint runs (README.md:48-55)."""
    fx = _load(golden_dir, "eval_b2_32x32x16")
    b, dhw = int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"])
    dt = torch.float64
    with torch.backends.mkldnn.flags(enabled=False):
        st1, _, _, _ = O.train_steps(O.fill_state(dt), [O.fill_batch(b, dhw, dtype=dt, seed=int(fx["meta/state_batch_seed"]))], 0, 1e-3, 240, 0)
        x = O.fill_batch(b, dhw, dtype=dt, seed=int(fx["meta/input_seed"]))[0]
        out, feats, masks = O.forward(st1, x, training=False)
    np.testing.assert_allclose(_samples(out, 512), fx["out/samples"], rtol=0, atol=1e-10)
    for i in range(3):
        np.testing.assert_allclose(feats[i][0].numpy(), fx[f"pro{i}"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(feats[i][1].numpy(), fx[f"pre{i}"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(_samples(masks[i], 512), fx[f"mask{i}/samples"], rtol=0, atol=1e-10)
    # eval mode leaves the state alone
    out2, _, _ = O.forward(st1, x, training=False)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("tag", ["v_elu", "v_prelu", "v_in", "v_inch3", "v_ncls2", "v_all"])
def test_constructor_variants_match_reference(tag, golden_dir):
    """The oracle built like PCRLv23d(act=..., norm=..., in_channels=..., n_class=...) against the REAL reference built with the same
    arguments (tests/golden/v_*.npz from make_golden.py --variants; models/pcrlv2_model_3d.py:15-16,22-25,98): forward outputs, the
    scalar O.variant_loss and the gradient of every parameter."""
    fx = _load(golden_dir, tag)
    kw = dict(n_class=int(fx["meta/n_class"]), in_channels=int(fx["meta/in_channels"]), act=str(fx["meta/act"]), norm=str(fx["meta/norm"]))
    dt = torch.float64
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    st0 = O.fill_state(dt, **kw)
    st = {k: (v.clone().requires_grad_(True) if not O.is_buffer(k) else v.clone()) for k, v in st0.items()}
    x = O.variant_input(int(fx["meta/b"]), tuple(int(v) for v in fx["meta/dhw"]), kw["in_channels"], dt)
    with torch.backends.mkldnn.flags(enabled=False):
        nb = {}
        out, feats, masks = O.forward(st, x, new_bufs=nb, act=kw["act"], norm=kw["norm"])
        L = O.variant_loss(out, feats, masks)
        pn = [k for k in st if not O.is_buffer(k)]
        grads = dict(zip(pn, torch.autograd.grad(L, [st[k] for k in pn], allow_unused=True)))
    assert abs(float(L) - float(fx["loss"])) < 1e-12
    np.testing.assert_allclose(_samples(out, 512), fx["fwd/out/samples"], rtol=0, atol=1e-10)
    for i in range(3):
        np.testing.assert_allclose(feats[i][0].detach().numpy(), fx[f"fwd/pro{i}"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(_samples(masks[i], 256), fx[f"fwd/mask{i}/samples"], rtol=0, atol=1e-10)
    for name, g in grads.items():
        if f"grad/{name}/none" in fx.files:
            assert g is None, name
            continue
        np.testing.assert_allclose(_samples(g, 64), fx[f"grad/{name}/samples"], rtol=0, atol=1e-9 * max(float(fx[f"grad/{name}/l2"]), 1e-3), err_msg=name)
    for key in fx.files:
        if key.startswith("buf1/"):
            np.testing.assert_allclose(nb[key[5:]].double().numpy(), fx[key], rtol=0, atol=1e-10, err_msg=key)


@pytest.mark.parametrize("tag", ["dp2_b4x2_32x32x16", "dp2chunk_b4x2_32x32x16"])
def test_data_parallel_semantics_match_reference(golden_dir, tag):
    """`oracle.train_steps_data_parallel` (nn.DataParallel, train_3d.py:54: per-replica statistics, replica 0's buffers persist, losses over
    the gathered batch, summed gradients, one SGD step) against the fixture `make_golden.py --data-parallel` produced with the REAL model."""
    fx = _load(golden_dir, tag)       # by-sample partition of the local views (the engine's default) / nn.DataParallel's literal [6B]-chunk scatter
    part = str(fx["meta/partition"]) if "meta/partition" in fx.files else "sample"
    b, dhw, nsteps, world = int(fx["meta/b_rank"]), tuple(int(v) for v in fx["meta/dhw"]), int(fx["meta/nsteps"]), int(fx["meta/world"])
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    st0 = O.fill_state(torch.float64)
    rb = [[O.fill_batch(b, dhw, dtype=torch.float64, seed=7 + 100 * s + 1000 * r) for r in range(world)] for s in range(nsteps)]
    with torch.backends.mkldnn.flags(enabled=False):
        # the first iteration only (the CPU suite's time budget): losses and the summed gradient; the state after both iterations is asserted
        # equal when the fixture is generated (make_golden.make_dp) and is what the two-rank GPU test is held to
        st, _mom, log, g0 = O.train_steps_data_parallel(st0, rb[:1], int(fx["meta/epoch"]), float(fx["meta/base_lr"]), int(fx["meta/epochs"]),
                                                        int(fx["meta/seed"]), partition=part)
    for s in range(1):
        for r in range(world):
            for k in ("loss", "loss1", "loss2", "loss4", "local_loss"):
                assert abs(log[s][r][k] - float(fx[f"step{s}/rank{r}/{k}"])) < 1e-10, (s, r, k)
            assert log[s][r]["index2"] == int(fx[f"step{s}/rank{r}/index2"])
    for name, g in g0.items():
        if f"grad/{name}/none" in fx.files:
            assert g is None, name
            continue
        ref = fx[f"grad/{name}/samples"]
        np.testing.assert_allclose(_samples(g, 64), ref, rtol=0, atol=1e-9 * max(np.abs(ref).max(), 1e-2))
