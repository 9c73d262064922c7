"""Input side of the 3D pre-training path: the reference's LUNA pre-task loader with the augmentations on the GPU.

Reference: `data.py:63-99` (DataGenerator.pcrlv2_luna_pretask), `datasets/lunaDataset.py:13-81` (Pcrlv2LunaPretask),
`utils.py:22-57` (file lists), `luna_preprocess.py:134-146` (what is on disk: `<series>_global_<k>.npy` = two overlapping
64x64x32 crops [2,64,64,32], `<series>_local_<k>.npy` = six 16^3 crops [6,16,16,16]).

The reference runs seven torchio transforms per crop in CPU DataLoader workers; at ~540 crops/s per GPU (8 views per crop) that
cannot keep one MI355X fed.  Here the workers only `np.load`; flips, affine resampling, blur, noise, gamma, patch swapping and
z-normalisation run batched on the device on the raw crops.

The transforms are hand-written gfx950 kernels (csrc/augment.hip, `pcrl_aug_*` in include/pcrl_hip.h); only the per-volume random
parameters are drawn with torch.

PARITY UNPINNED.  torchio is not installed in the build image and the reference holds no vectors for its augmentations, so these
are restatements of torchio's documented defaults (RandomFlip(axes=0, p=0.5); RandomAffine(scales 0.9-1.1, degrees +-10 per axis,
linear, pad with the image minimum); RandomBlur(std 0-2 per axis); RandomNoise(std 0-0.25); RandomGamma(log_gamma +-0.3);
RandomSwap(patch (8,4,4), 100 iterations); ZNormalization), tested against a float64 PyTorch restatement of the same definitions
(tests/aug_reference.py, tests/test_augment_gpu.py) and for their defining properties, not against torchio.  The batch contract -- (input1, input2, gt, gt2, [6 local views]), gt = the spatially transformed crop BEFORE the
intensity transforms (lunaDataset.py:37-41) -- is the reference's.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

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
# Batched device-side transforms on [B, D, H, W] float32 volumes: the random PARAMETERS (one set per volume) are drawn here with the
# torch generator; the transforms themselves are the hand-written kernels of csrc/augment.hip behind pcrl_aug_* (no torch arithmetic
# on the volumes).  tests/aug_reference.py restates the same definitions in float64 PyTorch for the parity tests.
# ---------------------------------------------------------------------------------------------------------------
BLUR_RADIUS = 8     # ceil(4 * max_std): torchio truncates its Gaussian at 4 sigma


def _u(gen, n, lo, hi, device):
    return lo + (hi - lo) * torch.rand(n, generator=gen, device=device)


def _rotation(deg):
    """[B,3] Euler angles in degrees (about the three spatial axes) -> [B,3,3] rotation matrices (x, then y, then z)."""
    a = deg * (math.pi / 180.0)
    c, s = torch.cos(a), torch.sin(a)
    one, zero = torch.ones_like(c[:, 0]), torch.zeros_like(c[:, 0])
    rx = torch.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1).view(-1, 3, 3)
    ry = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1).view(-1, 3, 3)
    rz = torch.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1).view(-1, 3, 3)
    return rz @ ry @ rx


def draw_spatial(gen, B, device, p_flip=0.5, scales=0.1, degrees=10.0):
    """RandomFlip(axes=0, p) + RandomAffine(scales, degrees) parameters -> (flip int32 [B], inv float32 [B,3,3]): `inv` maps an
    output voxel's centred coordinate to the input's (inverse of rotate . scale, (d,h,w) order, voxel units)."""
    flip = (torch.rand(B, generator=gen, device=device) < p_flip).to(torch.int32)
    sc = _u(gen, B * 3, 1.0 - scales, 1.0 + scales, device).view(B, 3)
    rot = _rotation(_u(gen, B * 3, -degrees, degrees, device).view(B, 3))
    # (rot . diag(sc))^-1 = diag(1 / sc) . rot^T in closed form: torch.linalg.inv on the device synchronises the host (it reads the LU
    # status back), which would stall the enqueue of the next training step behind the whole previous one
    inv = (rot.transpose(1, 2) / sc.unsqueeze(2)).to(torch.float32).contiguous()
    return flip, inv


_host_rngs: "collections.OrderedDict" = None      # id(generator) -> (generator, random.Random); the entry HOLDS the generator
_HOST_RNG_CAP = 64


def _host_rng_of(gen):
    """The host-side companion of a device generator (seeded from it once): scalars the kernels take BY VALUE -- the noise seed -- are drawn
    here, never read back from the device (an .item() between two training steps makes the host wait for the whole previous step).
    torch.Generator takes neither attributes nor weak references, so the table is keyed by id() and every entry keeps a strong reference to
    its generator: an id can then never be recycled for another generator while its entry lives (a stale entry would hand a NEW generator
    the old stream); the table is an LRU of 64 entries.  The companion's state is part of the augmentation stream: checkpoint it with
    the generator's own state (`host_rng_state` / `set_host_rng_state`)."""
    import collections
    import random
    global _host_rngs
    if _host_rngs is None:
        _host_rngs = collections.OrderedDict()
    ent = _host_rngs.get(id(gen))
    if ent is None or ent[0] is not gen:
        ent = _host_rngs[id(gen)] = (gen, random.Random(gen.initial_seed()))
        while len(_host_rngs) > _HOST_RNG_CAP:
            _host_rngs.popitem(last=False)
    else:
        _host_rngs.move_to_end(id(gen))
    return ent[1]


def host_rng_state(gen):
    """State of the generator's host companion (see _host_rng_of) -- save it next to `gen.get_state()` to reproduce an augmentation stream."""
    return _host_rng_of(gen).getstate()


def set_host_rng_state(gen, state):
    _host_rng_of(gen).setstate(state)


def draw_intensity(gen, B, device, max_blur=2.0, max_noise=0.25, log_gamma=0.3, host_rng=None):
    """RandomBlur / RandomNoise / RandomGamma parameters -> (sigma [3,B], noise_std [B], gamma [B], seed).  The per-volume parameters stay on
    the device; the noise seed (a kernel argument) comes from `host_rng` (default: the generator's host companion) -- no device read-back."""
    sigma = _u(gen, 3 * B, 0.0, max_blur, device).view(3, B).contiguous()
    noise_std = _u(gen, B, 0.0, max_noise, device)
    gamma = torch.exp(_u(gen, B, -log_gamma, log_gamma, device))
    seed = (host_rng if host_rng is not None else _host_rng_of(gen)).getrandbits(62)
    return sigma, noise_std, gamma, seed


def draw_swap(gen, B, dhw, device, patch=(8, 4, 4), iterations=100):
    """RandomSwap draws -> int32 [iterations, B, 2, 3] patch corners; a draw whose two patches overlap is skipped by torchio: both
    corners are set equal (an exchange with itself)."""
    D, H, W = dhw
    pd, ph, pw = patch
    r = torch.rand(iterations, B, 2, 3, generator=gen, device=device)
    o = (r * torch.tensor([D - pd + 1, H - ph + 1, W - pw + 1], device=device)).long()
    overlap = ((o[:, :, 0] - o[:, :, 1]).abs() < torch.tensor([pd, ph, pw], device=device)).all(dim=2)
    o = torch.where(overlap.view(iterations, B, 1, 1), o[:, :, :1].expand(-1, -1, 2, -1), o)
    return o.to(torch.int32).contiguous()


def _call(name, *args):
    from ._lib import lib, stream_handle
    lib().call(name, *args, stream_handle())


def apply_spatial(x, flip, inv):
    """flip + affine resampling of [B,D,H,W] float32 volumes on the device (pcrl_aug_volume_min, pcrl_aug_affine)."""
    B, D, H, W = x.shape
    x = x.contiguous()
    vmin = torch.empty(B, dtype=torch.float32, device=x.device)
    _call("pcrl_aug_volume_min", x, vmin, B, D * H * W)
    y = torch.empty_like(x)
    _call("pcrl_aug_affine", x, y, inv, flip, vmin, B, D, H, W)
    return y


def apply_intensity(x, sigma, noise_std, gamma, seed, swap_origins=None, patch=(8, 4, 4)):
    """blur (three separable passes) -> noise -> gamma -> [patch swaps] -> z-normalisation, all in csrc/augment.hip."""
    B, D, H, W = x.shape
    S = D * H * W
    a, b = x.contiguous(), torch.empty_like(x)
    for axis in range(3):
        _call("pcrl_aug_blur_axis", a, b, sigma[axis], B, D, H, W, axis, BLUR_RADIUS)
        a, b = b, (torch.empty_like(x) if axis == 0 else a)      # never write into the caller's tensor
    _call("pcrl_aug_noise_gamma", a, b, noise_std, gamma, B, S, seed)
    mean, rstd = torch.empty(B, dtype=torch.float32, device=x.device), torch.empty(B, dtype=torch.float32, device=x.device)
    _call("pcrl_aug_meanstd", b, mean, rstd, B, S)               # a permutation of the voxels leaves mean and std alone
    if swap_origins is not None:
        _call("pcrl_aug_swap", b, swap_origins, B, D, H, W, patch[0], patch[1], patch[2], swap_origins.shape[0])
    _call("pcrl_aug_znorm", b, a, mean, rstd, B, S)
    return a


class GpuLunaAugment:
    """data.py:73-89 as one batched device pass.  __call__(pair [B,2,D,H,W], locals [B,6,d,h,w]) -> the reference batch."""

    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GpuLunaAugment runs on the GPU (libpcrl_hip.so); there is no CPU fallback")
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        import random
        self.host_rng = random.Random(seed)       # by-value kernel arguments (the noise seed): drawn on the host, no device read-back

    def spatial(self, v):
        flip, inv = draw_spatial(self.gen, v.shape[0], self.device)
        return apply_spatial(v, flip, inv)

    def intensity(self, v, swap):
        sigma, noise_std, gamma, seed = draw_intensity(self.gen, v.shape[0], self.device, host_rng=self.host_rng)
        origins = draw_swap(self.gen, v.shape[0], tuple(v.shape[1:]), self.device) if swap else None
        return apply_intensity(v, sigma, noise_std, gamma, seed, origins)

    @torch.no_grad()
    def __call__(self, pair, local):
        pair = pair.to(self.device, torch.float32, non_blocking=True)
        local = local.to(self.device, torch.float32, non_blocking=True)
        B = pair.shape[0]
        views = self.spatial(pair.reshape((2 * B,) + tuple(pair.shape[2:])))                  # both global crops of every sample
        gt = views                                                                          # lunaDataset.py:37-38: before the intensity transforms (never written again)
        inp = self.intensity(views, swap=True)
        nl = local.shape[1]
        loc = self.intensity(self.spatial(local.reshape((nl * B,) + tuple(local.shape[2:]))), swap=False)
        g = lambda t, i: t.view((B, 2) + tuple(t.shape[1:]))[:, i].unsqueeze(1).contiguous()
        loc = loc.view((B, nl) + tuple(loc.shape[1:]))
        return g(inp, 0), g(inp, 1), g(gt, 0), g(gt, 1), [loc[:, i].unsqueeze(1).contiguous() for i in range(nl)]


class _SlotCrops(torch.utils.data.Dataset):
    """LunaCropPairs whose workers write a sample STRAIGHT into a slot of two persistent shared, page-locked batch buffers (key = (slot, row,
    file index)) and send back only (slot, row).  A DataLoader batch otherwise travels as freshly mapped shared memory: the consumer -- torch's
    pin_memory thread or a staging copy -- takes a page fault on every 4 KB of the 36 MB of a b = 32 batch (measured 10 ms per batch on an idle
    host, 22-38 ms next to a training loop that holds the GIL)."""

    def __init__(self, files, pair_buf, local_buf):
        self.files, self.pair, self.local = list(files), pair_buf, local_buf

    def __len__(self):
        return len(self.files)

    def __getitem__(self, key):
        slot, row, i = key
        name = self.files[i]
        self.pair[slot, row] = torch.from_numpy(np.load(name).astype(np.float32, copy=False))
        self.local[slot, row] = torch.from_numpy(np.load(name.replace("global", "local")).astype(np.float32, copy=False))     # lunaDataset.py:56
        return slot, row


class _SlotBatches(torch.utils.data.Sampler):
    """Batches of (slot, row, index) keys: a fresh permutation per epoch (DataLoader(shuffle=True), data.py:90), slots handed out round-robin."""

    def __init__(self, n, batch_size, shuffle, drop_last, nslots, seed):
        self.n, self.b, self.shuffle, self.drop_last, self.nslots = n, batch_size, shuffle, drop_last, nslots
        self.gen = torch.Generator().manual_seed(seed)
        self.counter = 0

    def __len__(self):
        return self.n // self.b if self.drop_last else (self.n + self.b - 1) // self.b

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.gen).tolist() if self.shuffle else list(range(self.n))
        for k in range(len(self)):
            idxs = order[k * self.b:(k + 1) * self.b]
            slot = self.counter % self.nslots
            self.counter += 1
            yield [(slot, row, i) for row, i in enumerate(idxs)]


def _pin_registered(t):
    """Page-lock an existing (shared-memory) host tensor in place (hipHostRegister); -> True when the runtime accepted it."""
    try:
        rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
        return int(rc) == 0 and t.is_pinned()
    except Exception:
        return False


AUG_STREAM = os.environ.get("PCRL_AUG_STREAM", "1") != "0"     # A/B switch: 0 = augment on the consumer's stream when the batch is asked for
# how a batch gets into page-locked memory for the asynchronous host-to-device copy:
#   slots  (default) workers write samples straight into persistent SHARED, page-locked batch slots (_SlotCrops): no copy at all in this process
#   ring   the DataLoader hands over shared-memory tensors; THIS loader copies them into a small ring of persistent pinned buffers
#   loader torch's pin_memory thread (a Python thread of the training process: it shares the GIL with the step's ~13 ms of enqueue work and
#          allocates fresh pinned memory per batch -- measured 10 ms per batch on an idle host, 38 ms next to a training loop)
#   none   pageable copies
PIN_MODE = os.environ.get("PCRL_LOADER_PIN", "slots")
LOADER_TIMING = os.environ.get("PCRL_LOADER_TIMING", "0") == "1"


def worker_affinity_init(worker_id: int):
    """DataLoader worker_init_fn: a worker forked from a rank whose launcher thread ddp.bind_rank_to_numa pinned inherits that narrow mask; the
    binding left the CPUs meant for the workers in $PCRL_WORKER_CPUS -- move there.  One CPU per worker ONLY where the rank's share was really
    split into launcher CPUs and worker CPUs ($PCRL_WORKER_CPUS_SPLIT=1, set by the binding) and there is a CPU per worker; an unsplit share
    (ddp.split_share returned the same list twice: too few CPUs) is the launcher's own CPUs -- the workers float over all of it, as they did
    before the split existed, instead of being hard-pinned onto the launcher thread's cores.  No variable, or a refusal: nothing happens."""
    cpus = [int(c) for c in os.environ.get("PCRL_WORKER_CPUS", "").split(",") if c.strip().isdigit()]
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return
    split = os.environ.get("PCRL_WORKER_CPUS_SPLIT", "0") == "1"
    try:
        info = torch.utils.data.get_worker_info()
        n = info.num_workers if info is not None else 0
        os.sched_setaffinity(0, [cpus[worker_id % len(cpus)]] if split and 0 < n <= len(cpus) else cpus)
    except OSError:
        pass


class AugmentedLoader:
    """DataLoader over raw crops + GpuLunaAugment: iterates batches with the contract of datasets/lunaDataset.py:79-81.

    One batch ahead: the host-to-device copy and the augmentation kernels of batch k + 1 are enqueued on their OWN stream BEFORE batch k is handed
    to the training loop, so they run under step k (~1.4 ms of HBM-bound kernels per b = 32 batch next to a ~31 ms step) instead of in
    front of step k + 1 on its critical chain; the consumer's stream waits for the batch's event when it receives it.  The reference's
    workers do the same job one batch ahead on the CPU (DataLoader prefetching, data.py:90-93)."""

    def __init__(self, files, batch_size, workers, device, shuffle=True, seed=0, drop_last=False):
        cuda = torch.device(device).type == "cuda"
        self.slots = None
        files = list(files)
        use_slots = cuda and PIN_MODE == "slots" and AUG_STREAM and workers > 0 and len(files) > 0
        if use_slots:
            prefetch = 2
            nslots = prefetch * workers + 6            # outstanding index batches + the batches this process still holds (see __iter__)
            pshape = tuple(np.load(files[0], mmap_mode="r").shape)
            lshape = tuple(np.load(files[0].replace("global", "local"), mmap_mode="r").shape)
            need = 4 * nslots * batch_size * (int(np.prod(pshape)) + int(np.prod(lshape)))
            try:        # the slots live in /dev/shm: a container with a small shm mount would die with SIGBUS on the first write, not with an exception
                vfs = os.statvfs("/dev/shm")
                use_slots = vfs.f_bavail * vfs.f_frsize > 1.25 * need
            except OSError:
                use_slots = False
            if not use_slots:
                print(f"[pcrlv2_amd.data] /dev/shm has no room for {need / 2**20:.0f} MB of batch slots: falling back to torch's pin_memory thread (slower, see PCRL_LOADER_PIN)")
        if use_slots:
            # zero-filled right after share_memory_(): the pages are COMMITTED now, so the /dev/shm room check of the next loader / rank sees real
            # usage instead of passing on sparse files and dying with SIGBUS later
            pair_buf = torch.empty((nslots, batch_size) + pshape, dtype=torch.float32).share_memory_().zero_()
            local_buf = torch.empty((nslots, batch_size) + lshape, dtype=torch.float32).share_memory_().zero_()
            pinned = _pin_registered(pair_buf) and _pin_registered(local_buf)
            if not pinned:
                # the copies out of pageable shared memory would be synchronous staged copies: the 841 -> 1015 crops/s of the slot loader is gone
                print("[pcrlv2_amd.data] warning: hipHostRegister refused the shared batch slots (locked-memory limit?): host-to-device copies are "
                      "staged and synchronous -- expect the loader to fall behind the step (PCRL_LOADER_PIN=ring copies into pinned buffers instead)", flush=True)
            self.slots = (pair_buf, local_buf, pinned)
            self.loader = torch.utils.data.DataLoader(_SlotCrops(files, pair_buf, local_buf), num_workers=workers, collate_fn=lambda items: (items[0][0], len(items)),
                                                      batch_sampler=_SlotBatches(len(files), batch_size, shuffle, drop_last, nslots, seed),
                                                      persistent_workers=True, prefetch_factor=prefetch, worker_init_fn=worker_affinity_init)
        else:
            self.loader = torch.utils.data.DataLoader(LunaCropPairs(files), batch_size=batch_size, shuffle=shuffle, num_workers=workers,
                                                      pin_memory=cuda and PIN_MODE in ("loader", "slots"), drop_last=drop_last,
                                                      persistent_workers=workers > 0, prefetch_factor=4 if workers > 0 else None,
                                                      worker_init_fn=worker_affinity_init if workers > 0 else None)
        self.augment = GpuLunaAugment(device, seed)
        self._stream = None
        self._ring, self._ring_pos = None, 0
        self._events = []
        self.timing = {"wait_loader_s": 0.0, "stage_s": 0.0, "enqueue_s": 0.0, "batches": 0}

    def __len__(self):
        return len(self.loader)

    @staticmethod
    def _tensors(batch):
        for t in batch:
            if torch.is_tensor(t):
                yield t
            elif t is not None:
                yield from t

    def _hand_over(self, batch, ev):
        cur = torch.cuda.current_stream(self.augment.device)
        cur.wait_event(ev)
        for t in self._tensors(batch):
            t.record_stream(cur)        # allocated on the augmentation stream, consumed on the training step's streams
        return batch

    def _stage(self, pair, local):
        """Shared-memory batch -> a slot of the pinned ring (three slots: the copy out of a slot finished two batches ago)."""
        if PIN_MODE != "ring" or pair.is_pinned():
            return pair, local, None
        if self._ring is None or self._ring[0][0].shape[1:] != pair.shape[1:] or self._ring[0][0].shape[0] < pair.shape[0]:
            self._ring = [[torch.empty(pair.shape, dtype=pair.dtype).pin_memory(), torch.empty(local.shape, dtype=local.dtype).pin_memory(), None] for _ in range(3)]
        slot = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        if slot[2] is not None:
            slot[2].synchronize()
        B = pair.shape[0]
        slot[0][:B].copy_(pair)
        slot[1][:B].copy_(local)
        return slot[0][:B], slot[1][:B], slot

    def close(self):
        """Release the page-lock on the shared batch slots (hipHostUnregister); the loader is unusable afterwards.  Host-to-device copies out of a
        slot may still be in flight (a consumer that broke out of the epoch; garbage collection): the loader's stream is drained first, and the
        DataLoader -- whose persistent workers still write into the slots -- is dropped before the buffers lose their page-lock."""
        if getattr(self, "_stream", None) is not None:
            try:
                self._stream.synchronize()
            except Exception:
                pass
        self._events = []
        if self.slots is not None:
            self.loader = None          # the iterator and its persistent workers go first
        if self.slots is not None and self.slots[2]:
            for t in self.slots[:2]:
                try:
                    torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
                except Exception:
                    pass
        self.slots = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __iter__(self):
        import time
        if not AUG_STREAM:
            for pair, local in self.loader:
                yield self.augment(pair, local)
            return
        dev = self.augment.device
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        ahead = None
        # A consumer that left the previous epoch early (a fixed step count with persistent workers) leaves index batches dispatched and slots
        # whose copies were never event-checked: drain this loader's stream and forget the old events before slots are handed out again
        self._stream.synchronize()
        self._events = []
        it = iter(self.loader)
        tm = self.timing
        while True:
            t0 = time.perf_counter()
            try:
                item = next(it)
            except StopIteration:
                break
            t1 = time.perf_counter()
            if self.slots is not None:
                # (slot, rows): the batch already lies in the shared page-locked buffers.  Receiving it made the DataLoader hand out the index
                # batch that will overwrite the slot of the batch six back: the copies out of that one are long done -- checked, not assumed
                sl, rows = item
                if len(self._events) >= 3:
                    self._events[-3].synchronize()
                pair, local, slot = self.slots[0][sl, :rows], self.slots[1][sl, :rows], None
            else:
                pair, local, slot = self._stage(*item)
            t2 = time.perf_counter()
            with torch.cuda.stream(self._stream):
                batch = self.augment(pair, local)
                ev = torch.cuda.Event()
                ev.record(self._stream)
                if slot is not None:
                    slot[2] = ev             # (behind the host-to-device copies of this slot)
            self._events = (self._events + [ev])[-4:]
            t3 = time.perf_counter()
            tm["wait_loader_s"] += t1 - t0
            tm["stage_s"] += t2 - t1
            tm["enqueue_s"] += t3 - t2
            tm["batches"] += 1
            if ahead is not None:
                yield self._hand_over(*ahead)
            ahead = (batch, ev)
        if ahead is not None:
            yield self._hand_over(*ahead)
        if LOADER_TIMING and tm["batches"]:
            n = tm["batches"]
            print("[loader] per batch: waiting for the DataLoader %.1f ms, staging into pinned memory %.1f ms, enqueue of copies + augmentation %.1f ms (pin mode %s, %d batches)"
                  % (1e3 * tm["wait_loader_s"] / n, 1e3 * tm["stage_s"] / n, 1e3 * tm["enqueue_s"] / n, PIN_MODE, n), flush=True)
            for k in ("wait_loader_s", "stage_s", "enqueue_s", "batches"):
                tm[k] = 0


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
