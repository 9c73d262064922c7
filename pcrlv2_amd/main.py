"""CLI of the MI355X engine -- drop-in for the reference's main.py (same flags, main.py:22-39).

    python main.py --data DIR --model pcrlv2 --b 32 --epochs 240 --lr 1e-3 --output saved_dir --n luna --d 3 --gpus 0 --ratio 1.0 --amp
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main.py ... --gpus 0,1,2,3,4,5,6,7      (one process per GPU)

Differences from the reference: `--gpus` selects the visible devices exactly as before, but multi-GPU runs use one
process per GPU (RCCL) instead of nn.DataParallel -- with a plain `python main.py` and several ids in --gpus the script
re-launches itself under torch.distributed.run.  `--momentum/--weight_decay` are parsed as floats.  `--d 2` runs the 2D
ResNet-18 U-Net path (pcrlv2_amd/train_2d.py; no segmentation_models_pytorch / torchvision needed) on `--data synthetic` only:
the chest X-ray input pipeline is not part of this engine.  `--data synthetic` trains on generated batches of the reference's
shapes (no dataset on disk needed); a LUNA pre-task directory is read by pcrlv2_amd/data.py (crops from disk, the reference's
torchio augmentations restated on the GPU -- parity with torchio unpinned).
"""
import argparse
import os
import subprocess
import sys
import warnings

warnings.filterwarnings('ignore')


# flag, default, type (None = store_true), help -- the reference's flags (main.py:22-39) with their defaults
_FLAGS = (
    ("data", "/data1/luchixiang/LUNA16/processed", str, "dataset directory, or 'synthetic'"),
    ("model", "pcrlv2", str, "model family"),
    ("phase", "pretask", str, "pretask | finetune | scratch"),
    ("b", 16, int, "batch size PER PROCESS"),
    ("epochs", 100, int, "last epoch index (inclusive)"),
    ("lr", 1e-3, float, "initial learning rate"),
    ("output", "./model_genesis_pretrain", str, "checkpoint directory"),
    ("n", "luna", str, "dataset name (goes into the checkpoint file name)"),
    ("d", 3, int, "2 or 3 dimensional model"),
    ("workers", 4, int, "loader workers"),
    ("gpus", "0,1,2,3", str, "visible device ids, comma separated"),
    ("ratio", 0.8, float, "fraction of the data used for pre-training"),
    ("momentum", 0.9, float, "SGD momentum"),
    ("weight_decay", 1e-4, float, "SGD weight decay"),
    ("seed", 42, int, "python/torch seed"),
    ("amp", False, None, "bfloat16 activations and MFMA operands"),
    ("steps_per_epoch", 16, int, "only with --data synthetic"),
    ("resume", "", str, "checkpoint to continue from (model, momentum buffers, epoch)"),
    ("encoder_weights", "", str, "only with --d 2: local ResNet-18 state_dict (torchvision key names) for the encoder; empty = random init "
                                 "(the reference's smp.Unet('resnet18') downloads ImageNet weights, which an offline engine cannot)"),
    ("size2d", 224, int, "only with --d 2 --data synthetic: side of the global views (locals are 96x96)"),
)


def build_parser():
    ap = argparse.ArgumentParser(description="PCRLv2 pre-training on MI355X")
    for name, default, kind, text in _FLAGS:
        if kind is None:
            ap.add_argument("--" + name, action="store_true", default=default, help=text)
        else:
            ap.add_argument("--" + name, default=default, type=kind, help=text)
    return ap


class SyntheticLunaLoader:
    """Batches with the contract of datasets/lunaDataset.py:79-81: (input1, input2, gt, gt2, [6 local views]).
    Generated on `device` (the GPU by default: at ~60 ms per b=32 step a CPU generator would be the bottleneck, as the reference's
    CPU augmentation workers are -- SURVEY 8f N3); `train_3d` accepts tensors on either side."""

    def __init__(self, b, steps, seed=0, device=None):
        import torch
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.b, self.steps, self.g = b, steps, torch.Generator(device=self.device).manual_seed(seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        import torch
        kw = dict(generator=self.g, device=self.device)
        for _ in range(self.steps):
            x1 = torch.randn(self.b, 1, 64, 64, 32, **kw)
            x2 = x1 + 0.1 * torch.randn(self.b, 1, 64, 64, 32, **kw)
            gt = torch.rand(self.b, 1, 64, 64, 32, **kw)
            # local views = 16^3 crops of the first global view + noise, as the real loader's are crops of the same volume (lunaDataset.py:60-78) and as
            # SURVEY 8(d) prescribes for loss-curve runs: with i.i.d. noise locals the local cosine term has no signal and its trajectory is chaotic
            # (round 5: bf16 and float32 runs of 2 000 steps ended 0.27 apart in that term alone, profiles/r05_long_run_compare_iid_locals.txt)
            loc = []
            for i in range(6):
                d0, h0, w0 = (i * 9) % 49, (i * 11) % 49, (i * 3) % 17
                loc.append(x1[:, :, d0:d0 + 16, h0:h0 + 16, w0:w0 + 16] + 0.1 * torch.randn(self.b, 1, 16, 16, 16, **kw))
            yield x1, x2, gt, gt, loc


class SyntheticChestLoader(SyntheticLunaLoader):
    """2D batches with the contract train_2d.py:133 consumes: (input1, input2, gt, gt2, [6 local views]) of [b,3,S,S] / [b,3,96,96]."""

    def __init__(self, b, steps, size, seed=0, device=None):
        super().__init__(b, steps, seed, device)
        self.size = size

    def __iter__(self):
        import torch
        kw = dict(generator=self.g, device=self.device)
        for _ in range(self.steps):
            x1 = torch.randn(self.b, 3, self.size, self.size, **kw)
            x2 = x1 + 0.1 * torch.randn(self.b, 3, self.size, self.size, **kw)
            gt = torch.rand(self.b, 3, self.size, self.size, **kw)
            loc = [torch.randn(self.b, 3, 96, 96, **kw) for _ in range(6)]
            yield x1, x2, gt, gt, loc


def get_dataloader(args):
    """`DataGenerator(args).pcrlv2_luna_pretask()` of the reference (data.py:63-99) -- `--data synthetic`: generated batches."""
    if args.data == 'synthetic' and args.d == 2:
        return {'train': SyntheticChestLoader(args.b, args.steps_per_epoch, args.size2d, args.seed + int(os.environ.get("RANK", "0"))), 'eval': None}
    if args.data == 'synthetic':
        return {'train': SyntheticLunaLoader(args.b, args.steps_per_epoch, args.seed + int(os.environ.get("RANK", "0"))), 'eval': None}
    if args.n == 'luna' and os.path.isdir(os.path.join(args.data, 'subset0')):
        from .data import luna_pretask_loaders     # raw .npy crops from disk, augmentations on the GPU (pcrlv2_amd/data.py)
        return luna_pretask_loaders(args)
    raise SystemExit("--data must be 'synthetic' or a LUNA pre-task directory (subset0..subset9 with <series>_global_<k>.npy / _local_<k>.npy, "
                     "luna_preprocess.py:134-146).  The chest X-ray (2D) pipeline is not part of this engine.")


def main(argv=None):
    args = build_parser().parse_args(argv)
    os.makedirs(args.output, exist_ok=True)
    ids = [g for g in args.gpus.split(',') if g != '']
    if len(ids) > 1 and "WORLD_SIZE" not in os.environ:
        env = dict(os.environ, HIP_VISIBLE_DEVICES=args.gpus, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={len(ids)}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(sys.argv[0])] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if "WORLD_SIZE" not in os.environ:
        os.environ["HIP_VISIBLE_DEVICES"] = args.gpus
    os.environ.setdefault("PCRL_LOADER_WORKERS", str(args.workers))     # ddp.bind_rank_to_numa keeps this many CPUs of the rank's share for the loader workers
    print(args)
    data_loader = get_dataloader(args)
    if args.model == 'pcrlv2' and args.phase == 'pretask' and args.d == 3:
        from .train_3d import train_pcrlv2_3d
        train_pcrlv2_3d(args, data_loader)
    elif args.model == 'pcrlv2' and args.phase == 'pretask' and args.d == 2:
        from .train_2d import train_pcrlv2
        train_pcrlv2(args, data_loader)


if __name__ == '__main__':
    main()
