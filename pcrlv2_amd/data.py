"""Input side of the 3D pre-training path: the reference's LUNA pre-task loader with the augmentations on the GPU.

Reference: `data.py:63-99` (DataGenerator.pcrlv2_luna_pretask), `datasets/lunaDataset.py:13-81` (Pcrlv2LunaPretask),
`utils.py:22-57` (file lists), `luna_preprocess.py:134-146` (what is on disk: `<series>_global_<k>.npy` = two overlapping
64x64x32 crops [2,64,64,32], `<series>_local_<k>.npy` = six 16^3 crops [6,16,16,16]).

The reference runs seven torchio transforms per crop in CPU DataLoader workers; at ~540 crops/s per GPU (8 views per crop) that
cannot keep one MI355X fed.  Here the workers only `np.load`; flips, affine resampling, blur, noise, gamma, patch swapping and
z-normalisation run batched on the device on the raw crops.

PARITY UNPINNED.  torchio is not installed in the build image and the reference holds no vectors for its augmentations, so these
are restatements of torchio's documented defaults (RandomFlip(axes=0, p=0.5); RandomAffine(scales 0.9-1.1, degrees +-10 per axis,
linear, pad with the image minimum); RandomBlur(std 0-2 per axis); RandomNoise(std 0-0.25); RandomGamma(log_gamma +-0.3);
RandomSwap(patch (8,4,4), 100 iterations); ZNormalization), tested for their defining properties (tests/test_data_cpu.py), not
against torchio.  The batch contract -- (input1, input2, gt, gt2, [6 local views]), gt = the spatially transformed crop BEFORE the
intensity transforms (lunaDataset.py:37-41) -- is the reference's.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn.functional as F

TRAIN_FOLDS, VALID_FOLDS = (0, 1, 2, 3, 4, 5, 6), (7, 8, 9)     # data.py:67-68


def luna_file_lists(data_dir: str, ratio: float, list_file: str = "train_val_txt/luna_train.txt"):
    """-> (x_train, x_valid): paths of the `_global_` files, as utils.get_luna_pretrain_list + get_luna_list select them:
    training = folds 0-6, only series whose id is in the first `ratio` of `list_file` (if that file exists: the reference
    requires it); validation = folds 7-9, every file."""
    keep = None
    if os.path.exists(list_file):
        with open(list_file) as f:
            names = [line.strip("\n") for line in f]
        keep = set(names[:int(len(names) * ratio)])

    def fold_files(fold, filt):
        d = os.path.join(data_dir, "subset" + str(fold))
        if not os.path.isdir(d):
            return []
        out = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "_global_" in f and (filt is None or f.split("_")[0] in filt)]
        return out

    x_train = [p for i in TRAIN_FOLDS for p in fold_files(i, keep)]
    x_valid = [p for i in VALID_FOLDS for p in fold_files(i, None)]
    return x_train, x_valid


class LunaCropPairs(torch.utils.data.Dataset):
    """Raw crops of one sample: (pair [2,64,64,32] float32, locals [6,16,16,16] float32).  No transforms here."""

    def __init__(self, files):
        self.files = list(files)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        name = self.files[i]
        pair = np.load(name).astype(np.float32)
        loc = np.load(name.replace("global", "local")).astype(np.float32)       # lunaDataset.py:56
        return torch.from_numpy(pair), torch.from_numpy(loc)


# ---------------------------------------------------------------------------------------------------------------
# Batched device-side transforms on [B, D, H, W] volumes (one random parameter set per volume)
# ---------------------------------------------------------------------------------------------------------------
def _u(gen, n, lo, hi, device):
    return lo + (hi - lo) * torch.rand(n, generator=gen, device=device)


def random_flip(x, gen, p=0.5):
    """torchio.RandomFlip(axes=0): mirror the first spatial axis with probability p."""
    flip = torch.rand(x.shape[0], generator=gen, device=x.device) < p
    return torch.where(flip.view(-1, 1, 1, 1), x.flip(1), x)


def _rotation(deg):
    """[B,3] Euler angles in degrees (about the three spatial axes) -> [B,3,3] rotation matrices (x, then y, then z)."""
    a = deg * (math.pi / 180.0)
    c, s = torch.cos(a), torch.sin(a)
    one, zero = torch.ones_like(c[:, 0]), torch.zeros_like(c[:, 0])
    rx = torch.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1).view(-1, 3, 3)
    ry = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1).view(-1, 3, 3)
    rz = torch.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1).view(-1, 3, 3)
    return rz @ ry @ rx


def random_affine(x, gen, scales=0.1, degrees=10.0):
    """torchio.RandomAffine defaults: per-axis scale U(1-s, 1+s), per-axis rotation U(-deg, deg) about the image centre, no
    translation, trilinear resampling, outside filled with the volume's minimum."""
    B = x.shape[0]
    sc = _u(gen, B * 3, 1.0 - scales, 1.0 + scales, x.device).view(B, 3)
    rot = _rotation(_u(gen, B * 3, -degrees, degrees, x.device).view(B, 3))
    # output voxel -> input coordinate: inverse of (rotate . scale); normalised coordinates of affine_grid are centred on the image
    fwd = rot @ torch.diag_embed(sc)
    inv = torch.linalg.inv(fwd)
    # physical (voxel) isotropy: normalised axes have different lengths, conjugate with the half-extents
    ext = torch.tensor([x.shape[3], x.shape[2], x.shape[1]], dtype=x.dtype, device=x.device) / 2.0     # affine_grid order: (W, H, D)
    perm = torch.tensor([2, 1, 0], device=x.device)                                                     # spatial (D,H,W) <-> grid (x=W,y=H,z=D)
    inv_g = inv[:, perm][:, :, perm]
    theta = (inv_g * ext.view(1, 1, 3)) / ext.view(1, 3, 1)
    theta = torch.cat([theta, torch.zeros(B, 3, 1, dtype=x.dtype, device=x.device)], dim=2)
    grid = F.affine_grid(theta, (B, 1) + tuple(x.shape[1:]), align_corners=False)
    lo = x.amin(dim=(1, 2, 3), keepdim=True)
    out = F.grid_sample((x - lo).unsqueeze(1), grid, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(1)
    return out + lo


def _gauss_kernels(sigma, radius):
    """[B] standard deviations (voxels) -> [B, 2*radius+1] normalised Gaussian taps (sigma -> 0: identity)."""
    t = torch.arange(-radius, radius + 1, device=sigma.device, dtype=sigma.dtype).view(1, -1)
    k = torch.exp(-0.5 * (t / sigma.clamp_min(1e-3).view(-1, 1)) ** 2)
    return k / k.sum(dim=1, keepdim=True)


def random_blur(x, gen, max_std=2.0):
    """torchio.RandomBlur: separable Gaussian, one std U(0, max_std) per axis and volume; symmetric ('reflect' in scipy) borders."""
    B = x.shape[0]
    radius = int(math.ceil(4.0 * max_std))
    out = x
    for axis in (1, 2, 3):
        k = _gauss_kernels(_u(gen, B, 0.0, max_std, x.device), radius)              # [B, K]
        n = out.shape[axis]
        r = min(radius, n)
        pad_lo = out.narrow(axis, 0, r).flip(axis)
        pad_hi = out.narrow(axis, n - r, r).flip(axis)
        padded = torch.cat([pad_lo, out, pad_hi], dim=axis)
        kk = k[:, radius - r: radius + r + 1]
        kk = kk / kk.sum(dim=1, keepdim=True)
        shape = [B, 1, 1, 1, 1]
        shape[axis + 1] = 2 * r + 1
        out = F.conv3d(padded.unsqueeze(0), kk.view(shape), groups=B).squeeze(0)     # batch folded into channels
    return out


def random_noise(x, gen, max_std=0.25):
    std = _u(gen, x.shape[0], 0.0, max_std, x.device).view(-1, 1, 1, 1)
    return x + std * torch.randn(x.shape, generator=gen, device=x.device, dtype=x.dtype)


def random_gamma(x, gen, log_gamma=0.3):
    g = torch.exp(_u(gen, x.shape[0], -log_gamma, log_gamma, x.device)).view(-1, 1, 1, 1)
    return torch.sign(x) * torch.abs(x) ** g          # torchio keeps the sign of negative intensities


def random_swap(x, gen, patch=(8, 4, 4), iterations=100):
    """torchio.RandomSwap: `iterations` times, exchange the contents of two random patches (per volume; a draw whose two patches
    overlap is skipped).  A permutation of the voxels: the multiset of intensities is unchanged."""
    B, D, H, W = x.shape
    pd, ph, pw = patch
    dev = x.device
    od, oh, ow = torch.meshgrid(torch.arange(pd, device=dev), torch.arange(ph, device=dev), torch.arange(pw, device=dev), indexing="ij")
    offs = (od * H + oh) * W + ow                                                     # [pd,ph,pw] flat offsets inside a volume
    flat = x.reshape(B, -1).clone()
    # all draws up front (one launch each); only the data-dependent gather/scatter chain stays sequential
    r = torch.rand(iterations, B, 2, 3, generator=gen, device=dev)
    o = (r * torch.tensor([D - pd + 1, H - ph + 1, W - pw + 1], device=dev)).long()               # [I,B,2,3] patch origins
    overlap = ((o[:, :, 0] - o[:, :, 1]).abs() < torch.tensor([pd, ph, pw], device=dev)).all(dim=2)     # overlapping pair: skip this swap
    o = torch.where(overlap.view(iterations, B, 1, 1), o[:, :, :1].expand(-1, -1, 2, -1), o)
    base = (o[..., 0] * H + o[..., 1]) * W + o[..., 2]                                              # [I,B,2]
    idx = base.unsqueeze(-1) + offs.view(1, 1, 1, -1)                                               # [I,B,2,P]
    for it in range(iterations):
        ia, ib = idx[it, :, 0], idx[it, :, 1]
        a, b = flat.gather(1, ia), flat.gather(1, ib)
        flat.scatter_(1, ia, b)                                                                      # as torchio: first <- second ...
        flat.scatter_(1, ib, a)                                                                      # ... second <- (old) first
    return flat.view(B, D, H, W)


def z_normalize(x):
    m = x.mean(dim=(1, 2, 3), keepdim=True)
    s = x.std(dim=(1, 2, 3), keepdim=True)            # unbiased, as torch.Tensor.std in torchio.ZNormalization
    return (x - m) / s.clamp_min(1e-12)


class GpuLunaAugment:
    """data.py:73-89 as one batched device pass.  __call__(pair [B,2,D,H,W], locals [B,6,d,h,w]) -> the reference batch."""

    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def spatial(self, v):
        return random_affine(random_flip(v, self.gen), self.gen)

    def intensity(self, v, swap):
        v = random_gamma(random_noise(random_blur(v, self.gen), self.gen), self.gen)
        if swap:
            v = random_swap(v, self.gen)
        return z_normalize(v)

    @torch.no_grad()
    def __call__(self, pair, local):
        pair = pair.to(self.device, torch.float32, non_blocking=True)
        local = local.to(self.device, torch.float32, non_blocking=True)
        B = pair.shape[0]
        views = self.spatial(pair.reshape((2 * B,) + tuple(pair.shape[2:])))                  # both global crops of every sample
        gt = views.clone()                                                                  # lunaDataset.py:37-38: before the intensity transforms
        inp = self.intensity(views, swap=True)
        nl = local.shape[1]
        loc = self.intensity(self.spatial(local.reshape((nl * B,) + tuple(local.shape[2:]))), swap=False)
        g = lambda t, i: t.view((B, 2) + tuple(t.shape[1:]))[:, i].unsqueeze(1).contiguous()
        loc = loc.view((B, nl) + tuple(loc.shape[1:]))
        return g(inp, 0), g(inp, 1), g(gt, 0), g(gt, 1), [loc[:, i].unsqueeze(1).contiguous() for i in range(nl)]


class AugmentedLoader:
    """DataLoader over raw crops + GpuLunaAugment: iterates batches with the contract of datasets/lunaDataset.py:79-81."""

    def __init__(self, files, batch_size, workers, device, shuffle=True, seed=0, drop_last=False):
        self.loader = torch.utils.data.DataLoader(LunaCropPairs(files), batch_size=batch_size, shuffle=shuffle, num_workers=workers,
                                                  pin_memory=torch.device(device).type == "cuda", drop_last=drop_last)
        self.augment = GpuLunaAugment(device, seed)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for pair, local in self.loader:
            yield self.augment(pair, local)


def luna_pretask_loaders(args, device=None):
    """`DataGenerator(args).pcrlv2_luna_pretask()` (data.py:63-99): {'train': ..., 'eval': ...}."""
    device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    x_train, x_valid = luna_file_lists(args.data, args.ratio)
    print(f"total train images {len(x_train)}, valid images {len(x_valid)}")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    seed = getattr(args, "seed", 0)
    # One process per GPU: every rank must run the SAME number of optimizer steps (each step ends in a collective), so the shards are
    # cut to equal length (like DistributedSampler's drop) and the ragged last batch is dropped; a single process keeps the
    # reference's loader (data.py:90-93: drop_last=False).
    if world > 1:
        x_train = x_train[:len(x_train) - len(x_train) % world]
    return {"train": AugmentedLoader(x_train[rank::world], args.b, args.workers, device, True, seed + rank, drop_last=world > 1),
            "eval": AugmentedLoader(x_valid, args.b, args.workers, device, False, seed)}
