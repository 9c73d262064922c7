"""GPU: the augmentation kernels (csrc/augment.hip) against the float64 PyTorch restatement of the same definitions
(tests/aug_reference.py) on identical drawn parameters, the batch contract of datasets/lunaDataset.py:79-81 and determinism.
Parity with torchio itself is UNPINNED (torchio is not installed; the reference holds no vectors)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import aug_reference as R  # noqa: E402
from pcrlv2_amd import data as D  # noqa: E402

DEV = "cuda"


def _vols(B, dhw, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((B,) + dhw, generator=g).to(DEV)


@pytest.mark.parametrize("dhw", [(64, 64, 32), (16, 16, 16), (12, 10, 6)])
def test_flip_affine_matches_definition(dhw):
    x = _vols(5, dhw, 1)
    gen = torch.Generator(device=DEV).manual_seed(3)
    flip, inv = D.draw_spatial(gen, 5, DEV)
    flip[0], flip[1] = 1, 0                                     # both branches
    y = D.apply_spatial(x, flip, inv)
    ref = R.ref_spatial(x, flip, inv)
    assert (y.double() - ref).abs().max().item() < 2e-5
    # identity parameters reproduce the input; a pure flip mirrors it
    eye = torch.eye(3, device=DEV).repeat(5, 1, 1).contiguous()
    zero = torch.zeros(5, dtype=torch.int32, device=DEV)
    assert torch.allclose(D.apply_spatial(x, zero, eye), x, atol=1e-6)
    assert torch.allclose(D.apply_spatial(x, zero + 1, eye), x.flip(1), atol=1e-6)
    # samples outside the volume take the volume's minimum: a strong zoom-out keeps min <= y <= max and hits the minimum in the corners
    out = D.apply_spatial(x, zero, (eye * 3.0).contiguous())
    assert torch.allclose(out[:, 0, 0, 0], x.amin(dim=(1, 2, 3)))


@pytest.mark.parametrize("dhw", [(64, 64, 32), (16, 16, 16), (6, 5, 4)])
def test_blur_gamma_swap_znorm_match_definition(dhw):
    B = 4
    x = _vols(B, dhw, 2) * 2.0 - 0.5                            # some negative intensities (RandomGamma keeps their sign)
    gen = torch.Generator(device=DEV).manual_seed(5)
    sigma, _noise, gamma, seed = D.draw_intensity(gen, B, DEV)
    sigma[0, 0] = 0.0                                           # sigma -> 0: identity along that axis
    patch = (8, 4, 4) if min(dhw) >= 8 else (2, 2, 2)
    origins = D.draw_swap(gen, B, dhw, DEV, patch=patch, iterations=25)
    zero = torch.zeros(B, device=DEV)
    got = D.apply_intensity(x, sigma, zero, gamma, seed, origins, patch=patch)       # noise std 0: deterministic part of the chain
    ref = R.ref_znorm(R.ref_swap(R.ref_gamma(R.ref_blur(x, sigma), gamma), origins.cpu(), patch))
    assert (got.double() - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item())
    # z-normalised: mean 0, unbiased std 1; the input tensor is untouched
    assert got.mean(dim=(1, 2, 3)).abs().max().item() < 1e-4 and (got.std(dim=(1, 2, 3)) - 1).abs().max().item() < 1e-4
    assert torch.equal(x, _vols(B, dhw, 2) * 2.0 - 0.5)
    # swaps alone are a permutation of the voxels
    one = torch.ones(B, device=DEV)
    y = x.clone()
    from pcrlv2_amd._lib import lib, stream_handle
    lib().call("pcrl_aug_swap", y, origins, B, dhw[0], dhw[1], dhw[2], patch[0], patch[1], patch[2], origins.shape[0], stream_handle())
    assert torch.equal(y.flatten(1).sort(dim=1).values, x.flatten(1).sort(dim=1).values) and not torch.equal(y, x)
    assert torch.equal(y.cpu(), R.ref_swap(x.cpu(), origins.cpu(), patch))


def test_noise_is_normal_with_the_drawn_std_and_keyed_by_seed():
    B, dhw = 3, (32, 32, 16)
    x = torch.zeros((B,) + dhw, device=DEV)
    std = torch.tensor([0.25, 0.1, 0.0], device=DEV)
    one = torch.ones(B, device=DEV)
    from pcrlv2_amd._lib import lib, stream_handle
    y, y2, y3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    S = x[0].numel()
    lib().call("pcrl_aug_noise_gamma", x, y, std, one, B, S, 1234, stream_handle())
    lib().call("pcrl_aug_noise_gamma", x, y2, std, one, B, S, 1234, stream_handle())
    lib().call("pcrl_aug_noise_gamma", x, y3, std, one, B, S, 1235, stream_handle())
    assert torch.equal(y, y2) and not torch.equal(y, y3)
    n = y[0] / 0.25
    assert abs(n.mean().item()) < 0.02 and abs(n.std().item() - 1) < 0.02 and abs((n ** 3).mean().item()) < 0.05 and abs((n ** 4).mean().item() - 3) < 0.15
    assert abs(y[1].std().item() - 0.1) < 2e-3 and y[2].abs().max().item() == 0.0
    assert abs(torch.corrcoef(torch.stack([y[0].flatten(), y[1].flatten()]))[0, 1].item()) < 0.02       # volumes get different streams


def test_batch_contract_and_determinism():
    aug = D.GpuLunaAugment(DEV, seed=7)
    pair, loc = torch.rand(3, 2, 64, 64, 32), torch.rand(3, 6, 16, 16, 16)
    x1, x2, g1, g2, locs = aug(pair, loc)
    assert x1.shape == x2.shape == g1.shape == g2.shape == (3, 1, 64, 64, 32)
    assert len(locs) == 6 and all(t.shape == (3, 1, 16, 16, 16) for t in locs)
    # inputs and locals are z-normalised per volume; the targets are NOT (they keep the crop's intensities)
    for t in (x1, x2, locs[0], locs[5]):
        assert t.mean(dim=(1, 2, 3, 4)).abs().max().item() < 1e-4 and (t.std(dim=(1, 2, 3, 4)) - 1).abs().max().item() < 1e-3
    assert g1.min().item() >= -1e-5 and g1.max().item() <= 1 + 1e-5
    again = D.GpuLunaAugment(DEV, seed=7)(pair, loc)
    assert torch.equal(again[0], x1) and torch.equal(again[4][3], locs[3])
    other = D.GpuLunaAugment(DEV, seed=8)(pair, loc)
    assert not torch.equal(other[0], x1)
    with pytest.raises(RuntimeError):
        D.GpuLunaAugment("cpu")


def test_loader_end_to_end(tmp_path, monkeypatch):
    import types
    from test_data_cpu import _make_tree
    _make_tree(tmp_path, series_per_fold=1, pairs=1)
    args = types.SimpleNamespace(data=str(tmp_path), ratio=1.0, b=3, workers=0, seed=0)
    monkeypatch.chdir(tmp_path)        # no train_val_txt/luna_train.txt here: every series is kept
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    loaders = D.luna_pretask_loaders(args, device=DEV)
    assert len(loaders["train"]) == 3 and len(loaders["eval"]) == 1           # 7 files in batches of 3; 3 validation files
    batch = next(iter(loaders["train"]))
    assert batch[0].is_cuda and batch[0].shape == (3, 1, 64, 64, 32) and batch[2].shape == (3, 1, 64, 64, 32) and len(batch[4]) == 6
    assert batch[4][0].shape == (3, 1, 16, 16, 16)
