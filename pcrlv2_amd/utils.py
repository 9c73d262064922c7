"""Host-side helpers of the training loop (reference: utils.py:101-137)."""
import math


def adjust_learning_rate(epoch, args, optimizer):
    """Per-epoch cosine schedule, utils.py:101-114: lr = args.lr * 0.5 * (1 + cos(pi * epoch / args.epochs))."""
    lr = args.lr * 0.5 * (1. + math.cos(math.pi * epoch / args.epochs))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


class AverageMeter(object):
    """Running value / average, utils.py:117-137 (`val`, `avg`, `sum`, `count`).  Values and weights may be python numbers or 0-d device
    tensors.  Device values are only QUEUED by update(): the running sum is formed when `avg` / `sum` / `count` is read (one stack + one
    weighted sum for everything queued), so a training step launches nothing for its meters and never synchronises the GPU."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self._sum = 0
        self._count = 0
        self._queue = []

    def update(self, val, n=1):
        self.val = val
        if hasattr(val, "is_cuda") or hasattr(n, "is_cuda"):
            self._queue.append((val, n))
        else:
            self._sum = self._sum + val * n
            self._count += n

    def _settle(self):
        if self._queue:
            import torch
            dev = next(t.device for pair in self._queue for t in pair if hasattr(t, "is_cuda"))
            as_t = lambda v: v.detach().reshape(()).to(torch.float32) if hasattr(v, "is_cuda") else torch.tensor(float(v), device=dev)
            vals = torch.stack([as_t(v) for v, _ in self._queue])
            ns = torch.stack([as_t(n) for _, n in self._queue])
            self._sum = self._sum + (vals * ns).sum()
            self._count = self._count + ns.sum()
            self._queue = []

    @property
    def sum(self):
        self._settle()
        return self._sum

    @property
    def count(self):
        self._settle()
        return self._count

    @property
    def avg(self):
        self._settle()
        if isinstance(self._count, (int, float)):
            return self._sum / self._count if self._count else 0
        import torch        # device count: an epoch whose every step was skipped by the divergence guard has weight 0 -- report 0 like an empty meter, not 0/0
        return torch.where(self._count > 0, self._sum / self._count.clamp_min(1e-30), torch.zeros_like(self._sum))
