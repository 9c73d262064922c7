"""Host-side helpers of the training loop (reference: utils.py:101-137)."""
import math


def adjust_learning_rate(epoch, args, optimizer):
    """Per-epoch cosine schedule, utils.py:101-114: lr = args.lr * 0.5 * (1 + cos(pi * epoch / args.epochs))."""
    lr = args.lr * 0.5 * (1. + math.cos(math.pi * epoch / args.epochs))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


class AverageMeter(object):
    """Running value / average, utils.py:117-137.  Values may be python numbers or 0-d device tensors;
    tensors are only converted when read (`float(meter.val)`), so updating never synchronises the GPU."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count
