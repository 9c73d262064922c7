"""CLI of the MI355X engine -- drop-in for the reference's main.py (same flags, main.py:22-39).

    python main.py --data DIR --model pcrlv2 --b 32 --epochs 240 --lr 1e-3 --output saved_dir --n luna --d 3 --gpus 0 --ratio 1.0 --amp
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main.py ... --gpus 0,1,2,3,4,5,6,7      (one process per GPU)

Differences from the reference: `--gpus` selects the visible devices exactly as before, but multi-GPU runs use one
process per GPU (RCCL) instead of nn.DataParallel -- with a plain `python main.py` and several ids in --gpus the script
re-launches itself under torch.distributed.run.  `--momentum/--weight_decay` are parsed as floats.  `--d 2` (the 2D
ResNet18 path, needs segmentation_models_pytorch) is not part of this engine yet.  `--data synthetic` trains on
generated LUNA-shaped batches (no dataset on disk needed).
"""
import argparse
import os
import subprocess
import sys
import warnings

warnings.filterwarnings('ignore')


def build_parser():
    parser = argparse.ArgumentParser(description='Self Training benchmark')
    parser.add_argument('--data', metavar='DIR', default='/data1/luchixiang/LUNA16/processed', help='path to dataset')
    parser.add_argument('--model', metavar='MODEL', default='pcrlv2', help='choose the model')
    parser.add_argument('--phase', default='pretask', type=str, help='pretask or finetune or train from scratch')
    parser.add_argument('--b', default=16, type=int, help='batch size (per process)')
    parser.add_argument('--epochs', default=100, type=int, help='epochs to train')
    parser.add_argument('--lr', default=1e-3, type=float, help='learning rate')
    parser.add_argument('--output', default='./model_genesis_pretrain', type=str, help='output path')
    parser.add_argument('--n', default='luna', type=str, help='dataset to use')
    parser.add_argument('--d', default=3, type=int, help='3d or 2d to run')
    parser.add_argument('--workers', default=4, type=int, help='num of workers')
    parser.add_argument('--gpus', default='0,1,2,3', type=str, help='gpu indexs')
    parser.add_argument('--ratio', default=0.8, type=float, help='ratio of data used for pretraining')
    parser.add_argument('--momentum', default=0.9, type=float)
    parser.add_argument('--weight_decay', default=1e-4, type=float)
    parser.add_argument('--seed', default=42, type=int)
    parser.add_argument('--amp', action='store_true', default=False)
    parser.add_argument('--steps_per_epoch', default=16, type=int, help='only with --data synthetic')
    return parser


class SyntheticLunaLoader:
    """Batches with the contract of datasets/lunaDataset.py:79-81: (input1, input2, gt, gt2, [6 local views])."""

    def __init__(self, b, steps, seed=0):
        import torch
        self.b, self.steps, self.g = b, steps, torch.Generator().manual_seed(seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        import torch
        for _ in range(self.steps):
            x1 = torch.randn(self.b, 1, 64, 64, 32, generator=self.g)
            x2 = x1 + 0.1 * torch.randn(self.b, 1, 64, 64, 32, generator=self.g)
            gt = torch.rand(self.b, 1, 64, 64, 32, generator=self.g)
            loc = [torch.randn(self.b, 1, 16, 16, 16, generator=self.g) for _ in range(6)]
            yield x1, x2, gt, gt, loc


def get_dataloader(args):
    if args.data == 'synthetic':
        return {'train': SyntheticLunaLoader(args.b, args.steps_per_epoch, args.seed + int(os.environ.get("RANK", "0"))), 'eval': None}
    raise SystemExit("The LUNA/chest data pipelines (reference data.py, torchio/torchvision) are host-side code outside this engine: "
                     "pass --data synthetic, or build the loaders with the reference's data.DataGenerator and call "
                     "pcrlv2_amd.train_3d.train_pcrlv2_3d(args, {'train': loader}) directly (same batch contract).")


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
    print(args)
    data_loader = get_dataloader(args)
    if args.model == 'pcrlv2' and args.phase == 'pretask' and args.d == 3:
        from .train_3d import train_pcrlv2_3d
        train_pcrlv2_3d(args, data_loader)
    elif args.d == 2:
        raise SystemExit("--d 2 (PCRLv2 ResNet18 / segmentation_models_pytorch) is not built in this engine yet (SURVEY 8f N1)")


if __name__ == '__main__':
    main()
