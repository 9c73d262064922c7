"""Input pipeline, CPU: file selection as the reference's utils.get_luna_list, the batch contract of
datasets/lunaDataset.py:79-81, and the DEFINING PROPERTIES of the device-side augmentations.  Parity with torchio itself is
unpinned (torchio is not installed here and the reference holds no vectors): see pcrlv2_amd/data.py."""
import os
import types

import numpy as np
import torch

from pcrlv2_amd import data as D


def _make_tree(root, series_per_fold=2, pairs=2):
    names = []
    for fold in range(10):
        d = root / f"subset{fold}"
        d.mkdir()
        for s in range(series_per_fold):
            name = f"s{fold}x{s}"
            names.append(name)
            for k in range(pairs):
                np.save(d / f"{name}_global_{k}.npy", np.random.rand(2, 64, 64, 32).astype(np.float32))
                np.save(d / f"{name}_local_{k}.npy", np.random.rand(6, 16, 16, 16).astype(np.float32))
    return names


def test_file_lists_follow_the_reference_folds(tmp_path):
    names = _make_tree(tmp_path)
    lst = tmp_path / "luna_train.txt"
    lst.write_text("\n".join(names) + "\n")
    x_train, x_valid = D.luna_file_lists(str(tmp_path), 1.0, str(lst))
    assert len(x_train) == 7 * 2 * 2 and len(x_valid) == 3 * 2 * 2          # folds 0-6 / 7-9, every `_global_` file
    assert all("_global_" in p for p in x_train + x_valid)
    half, _ = D.luna_file_lists(str(tmp_path), 0.5, str(lst))               # first half of the list = series of folds 0..4
    assert {os.path.basename(p).split("_")[0] for p in half} == set(names[:10]) & {os.path.basename(p).split("_")[0] for p in x_train}
    ds = D.LunaCropPairs(x_train)
    pair, loc = ds[3]
    assert pair.shape == (2, 64, 64, 32) and loc.shape == (6, 16, 16, 16) and pair.dtype == torch.float32


def test_batch_contract_and_determinism():
    aug = D.GpuLunaAugment("cpu", seed=7)
    pair, loc = torch.rand(3, 2, 32, 32, 16), torch.rand(3, 6, 16, 16, 16)
    x1, x2, g1, g2, locs = aug(pair, loc)
    assert x1.shape == x2.shape == g1.shape == g2.shape == (3, 1, 32, 32, 16)
    assert len(locs) == 6 and all(t.shape == (3, 1, 16, 16, 16) for t in locs)
    # inputs and locals are z-normalised per volume; the targets are NOT (they keep the crop's intensities)
    for t in (x1, x2, locs[0], locs[5]):
        assert torch.allclose(t.mean(dim=(1, 2, 3, 4)), torch.zeros(3), atol=1e-4) and torch.allclose(t.std(dim=(1, 2, 3, 4)), torch.ones(3), atol=1e-3)
    assert g1.min() >= -1e-5 and g1.max() <= 1 + 1e-5
    again = D.GpuLunaAugment("cpu", seed=7)(pair, loc)
    assert torch.equal(again[0], x1) and torch.equal(again[4][3], locs[3])
    other = D.GpuLunaAugment("cpu", seed=8)(pair, loc)
    assert not torch.equal(other[0], x1)


def test_transform_properties():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(8, 16, 12, 10, generator=g)
    f = D.random_flip(x, g)
    assert all(torch.equal(f[i], x[i]) or torch.equal(f[i], x[i].flip(0)) for i in range(8))
    assert any(torch.equal(f[i], x[i].flip(0)) for i in range(8)) and any(torch.equal(f[i], x[i]) for i in range(8))
    assert torch.allclose(D.random_affine(x, g, scales=0.0, degrees=0.0), x, atol=1e-5)            # identity parameters
    a = D.random_affine(x, g)
    assert a.shape == x.shape and (a.amin(dim=(1, 2, 3)) >= x.amin(dim=(1, 2, 3)) - 1e-5).all()      # padded with the volume minimum
    assert torch.allclose(D.random_blur(x, g, max_std=1e-6), x, atol=1e-5)                           # sigma -> 0: identity
    b = D.random_blur(x, g)
    assert (b.var(dim=(1, 2, 3)) < x.var(dim=(1, 2, 3))).all() and torch.allclose(b.mean(dim=(1, 2, 3)), x.mean(dim=(1, 2, 3)), atol=2e-2)
    n = D.random_noise(torch.zeros(64, 8, 8, 8), g)
    assert (n.std(dim=(1, 2, 3)) <= 0.25 * 1.2).all() and n.std(dim=(1, 2, 3)).max() > 0.1
    assert torch.allclose(D.random_gamma(x, g, log_gamma=0.0), x)
    y = D.random_gamma(x - 0.5, g)
    assert torch.equal(torch.sign(y), torch.sign(x - 0.5))
    s = D.random_swap(x, g, patch=(4, 2, 2), iterations=20)
    assert not torch.equal(s, x)
    assert torch.equal(s.reshape(8, -1).sort(dim=1).values, x.reshape(8, -1).sort(dim=1).values)     # a permutation of the voxels
    z = D.z_normalize(x * 3 + 2)
    assert torch.allclose(z.mean(dim=(1, 2, 3)), torch.zeros(8), atol=1e-5) and torch.allclose(z.std(dim=(1, 2, 3)), torch.ones(8), atol=1e-5)


def test_rotation_is_a_rotation_about_the_centre():
    # 90 degrees about the first spatial axis maps the (h, w) plane onto itself: compare with torch.rot90 on a cube
    x = torch.rand(2, 8, 8, 8)
    deg = torch.tensor([[90.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    r = D._rotation(deg)
    assert torch.allclose(r @ r.transpose(1, 2), torch.eye(3).expand(2, 3, 3), atol=1e-6) and torch.allclose(torch.linalg.det(r), torch.ones(2), atol=1e-6)
    assert torch.allclose(r[1], torch.eye(3), atol=1e-7)


def test_loader_end_to_end(tmp_path):
    names = _make_tree(tmp_path, series_per_fold=1, pairs=1)
    args = types.SimpleNamespace(data=str(tmp_path), ratio=1.0, b=3, workers=0, seed=0)
    cwd = os.getcwd()
    os.chdir(tmp_path)        # no train_val_txt/luna_train.txt here: every series is kept
    try:
        loaders = D.luna_pretask_loaders(args, device="cpu")
    finally:
        os.chdir(cwd)
    assert len(loaders["train"]) == 3 and len(loaders["eval"]) == 1           # 7 files in batches of 3; 3 validation files
    batch = next(iter(loaders["train"]))
    assert batch[0].shape == (3, 1, 64, 64, 32) and batch[2].shape == (3, 1, 64, 64, 32) and len(batch[4]) == 6
    assert batch[4][0].shape == (3, 1, 16, 16, 16)
