"""Input pipeline, CPU: file selection as the reference's utils.get_luna_list, the batch contract of
datasets/lunaDataset.py:79-81, the host-side parameter draws, and the DEFINING PROPERTIES of the float64 restatement the augmentation kernels are tested
against on the GPU (tests/test_augment_gpu.py).  Parity with torchio itself is unpinned (torchio is not installed here and the reference
holds no vectors): see pcrlv2_amd/data.py."""
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


def test_parameter_draws_follow_the_torchio_defaults():
    """The random PARAMETERS are host-side (torch); the transforms run in csrc/augment.hip (tests/test_augment_gpu.py)."""
    g = torch.Generator().manual_seed(1)
    flip, inv = D.draw_spatial(g, 2000, "cpu")
    assert flip.dtype == torch.int32 and 0.45 < flip.float().mean() < 0.55                             # RandomFlip(p=0.5)
    fwd = torch.linalg.inv(inv)                                                                        # rotate . scale
    sc = torch.linalg.svdvals(fwd)
    assert sc.min() >= 0.9 - 1e-5 and sc.max() <= 1.1 + 1e-5                                           # scales U(0.9, 1.1)
    sigma, nstd, gamma, seed = D.draw_intensity(g, 2000, "cpu")
    assert sigma.shape == (3, 2000) and 0 <= sigma.min() and sigma.max() <= 2.0 and 0.9 < sigma.mean() < 1.1
    assert nstd.min() >= 0 and nstd.max() <= 0.25 and gamma.min() >= np.exp(-0.3) - 1e-6 and gamma.max() <= np.exp(0.3) + 1e-6
    o = D.draw_swap(g, 64, (64, 64, 32), "cpu")
    assert o.shape == (100, 64, 2, 3) and o.dtype == torch.int32
    assert (o >= 0).all() and (o[..., 0] <= 64 - 8).all() and (o[..., 1] <= 64 - 4).all() and (o[..., 2] <= 32 - 4).all()
    d = (o[:, :, 0] - o[:, :, 1]).abs()
    apart = (d >= torch.tensor([8, 4, 4])).any(dim=2)
    same = (d == 0).all(dim=2)
    assert (apart | same).all() and apart.float().mean() > 0.9                                         # overlapping draws are skipped
    # same seed -> same draws
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    assert torch.equal(D.draw_spatial(g1, 8, "cpu")[1], D.draw_spatial(g2, 8, "cpu")[1])


def test_reference_restatement_properties():
    """Defining properties of the float64 restatement the GPU kernels are tested against (tests/aug_reference.py)."""
    import aug_reference as R
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4, 16, 12, 10, generator=g)
    eye, zero = torch.eye(3).repeat(4, 1, 1), torch.zeros(4, dtype=torch.int32)
    assert torch.allclose(R.ref_spatial(x, zero, eye), x.double(), atol=1e-12)
    assert torch.allclose(R.ref_spatial(x, zero + 1, eye), x.flip(1).double(), atol=1e-12)
    flip, inv = D.draw_spatial(g, 4, "cpu")
    a = R.ref_spatial(x, flip, inv)
    assert (a.amin(dim=(1, 2, 3)) >= x.amin(dim=(1, 2, 3)) - 1e-9).all() and (a.amax(dim=(1, 2, 3)) <= x.amax(dim=(1, 2, 3)) + 1e-9).all()
    sig = torch.full((3, 4), 1e-6)
    assert torch.allclose(R.ref_blur(x, sig), x.double(), atol=1e-9)                                    # sigma -> 0: identity
    b = R.ref_blur(x, torch.full((3, 4), 1.5))
    assert (b.var(dim=(1, 2, 3)) < x.var(dim=(1, 2, 3))).all() and torch.allclose(b.mean(dim=(1, 2, 3)), x.double().mean(dim=(1, 2, 3)), atol=2e-2)
    assert torch.allclose(R.ref_gamma(x, torch.ones(4)), x.double())
    assert torch.equal(torch.sign(R.ref_gamma(x - 0.5, torch.full((4,), 1.2))), torch.sign(x - 0.5).double())
    o = D.draw_swap(g, 4, (16, 12, 10), "cpu", patch=(4, 2, 2), iterations=20)
    s2 = R.ref_swap(x, o, patch=(4, 2, 2))
    assert not torch.equal(s2, x) and torch.equal(s2.reshape(4, -1).sort(dim=1).values, x.reshape(4, -1).sort(dim=1).values)
    z = R.ref_znorm(x * 3 + 2)
    assert torch.allclose(z.mean(dim=(1, 2, 3)), torch.zeros(4, dtype=torch.float64), atol=1e-9) and torch.allclose(z.std(dim=(1, 2, 3)), torch.ones(4, dtype=torch.float64), atol=1e-9)


def test_rotation_is_a_rotation_about_the_centre():
    deg = torch.tensor([[90.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    r = D._rotation(deg)
    assert torch.allclose(r @ r.transpose(1, 2), torch.eye(3).expand(2, 3, 3), atol=1e-6) and torch.allclose(torch.linalg.det(r), torch.ones(2), atol=1e-6)
    assert torch.allclose(r[1], torch.eye(3), atol=1e-7)


def test_loader_shards_have_equal_length(tmp_path, monkeypatch):
    """One process per GPU: every rank must see the same number of batches (each step ends in a collective)."""
    _make_tree(tmp_path, series_per_fold=1, pairs=1)          # 7 training files
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(D, "AugmentedLoader", lambda files, b, workers, device, shuffle=True, seed=0, drop_last=False: (list(files), drop_last))
    lens = []
    for rank in range(2):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "2")
        args = types.SimpleNamespace(data=str(tmp_path), ratio=1.0, b=2, workers=0, seed=0)
        files, drop = D.luna_pretask_loaders(args, device="cpu")["train"]
        lens.append(len(files))
        assert drop is True
    assert lens == [3, 3]                                      # 7 files -> 6, three per rank
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    files, drop = D.luna_pretask_loaders(types.SimpleNamespace(data=str(tmp_path), ratio=1.0, b=2, workers=0, seed=0), device="cpu")["train"]
    assert len(files) == 7 and drop is False                   # single process: the reference's loader (drop_last=False)


def test_slot_loader_workers_write_the_right_samples_into_the_shared_buffers(tmp_path):
    """data._SlotCrops / _SlotBatches (the default loader form on the GPU): worker processes write each sample into (slot, row) of the shared
    batch buffers and send back (slot, rows).  Every batch read back from its slot holds exactly the files the sampler assigned, every file
    is delivered once per epoch, slots rotate, the last batch is ragged (drop_last=False, data.py:90-93) -- over two epochs with two workers."""
    _make_tree(tmp_path, series_per_fold=3, pairs=1)
    files, _ = D.luna_file_lists(str(tmp_path), 1.0, str(tmp_path / "no_list.txt"))
    assert len(files) == 21
    b, nslots = 4, 2 * 2 + 6
    pshape = tuple(np.load(files[0]).shape)
    lshape = tuple(np.load(files[0].replace("global", "local")).shape)
    pair_buf = torch.zeros((nslots, b) + pshape).share_memory_()
    local_buf = torch.zeros((nslots, b) + lshape).share_memory_()
    sampler = D._SlotBatches(len(files), b, True, False, nslots, seed=5)
    loader = torch.utils.data.DataLoader(D._SlotCrops(files, pair_buf, local_buf), num_workers=2, collate_fn=lambda items: (items[0][0], len(items)),
                                         batch_sampler=sampler, persistent_workers=True, prefetch_factor=2)
    seen_slots = []
    for epoch in range(2):
        # the sampler's plan of this epoch, replayed from a copy of its generator state
        g = torch.Generator()
        g.set_state(sampler.gen.get_state())
        order = torch.randperm(len(files), generator=g).tolist()
        delivered = []
        for k, (slot, rows) in enumerate(loader):
            idxs = order[k * b:(k + 1) * b]
            assert rows == len(idxs)
            for r, i in enumerate(idxs):
                assert torch.equal(pair_buf[slot, r], torch.from_numpy(np.load(files[i])))
                assert torch.equal(local_buf[slot, r], torch.from_numpy(np.load(files[i].replace("global", "local"))))
            delivered += idxs
            seen_slots.append(slot)
        assert sorted(delivered) == list(range(len(files)))
    assert seen_slots == [k % nslots for k in range(len(seen_slots))]
    assert len(seen_slots) == 2 * 6 and seen_slots[5] == 5      # 21 files, b = 4: five full batches and a ragged one per epoch
