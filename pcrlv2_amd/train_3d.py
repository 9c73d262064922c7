"""3D pre-training loop on the MI355X engine -- drop-in for the reference's train_3d.py.

Same entry point `train_pcrlv2_3d(args, data_loader, out_channel=3)`, same `cos_loss`, same loss
assembly, LR schedule, log line, checkpoint dict and file name.  Differences, all deliberate:

  * compute: PCRLv23d runs on hand-written gfx950 kernels (pcrlv2_amd.models); losses and SGD too;
  * `--amp` (apex O1 fp16 in the reference, train_3d.py:52-53) selects bfloat16 activations / MFMA operands
    with float32 accumulation, statistics and master weights; no loss scaling is needed with bf16;
  * `nn.DataParallel` (train_3d.py:54) is replaced by one process per GPU + RCCL all-reduce
    (`pcrlv2_amd.ddp`): launch with torchrun, or plain `python main.py ...` for one GPU.  `--b` is the
    PER-PROCESS batch here (the reference splits a global batch over replicas);
  * meters hold device scalars and are read only when the log line is printed, so a step does not
    synchronise the GPU (the reference calls .item() twice and cuda.synchronize() every iteration);
  * the divergence guard (train_3d.py:140-142) is evaluated as `epoch > 10 and loss > 1000` so the
    device->host read happens only when the reference would act on it;
  * `--seed`, ignored by the reference, seeds python `random` (scale draws) and torch.
"""
from __future__ import print_function

import math
import os
import random
import sys
import time

import torch

from . import ddp as _ddp
from .functions import cosine_mean, mse_loss
from .models import PCRLv23d
from .optim import FusedSGD
from .utils import AverageMeter, adjust_learning_rate


class MSELoss:
    """criterion of train_3d.py:56 on the fused sigmoid-map MSE kernel."""

    def cuda(self):
        return self

    def __call__(self, p, gt):
        return mse_loss(p, gt)


class CosineSimilarityMean:
    """train_3d.py:57's nn.CosineSimilarity(), fused with the `.mean()` every call site applies."""
    returns_mean = True

    def cuda(self):
        return self

    def __call__(self, x, y):
        return cosine_mean(x, y)


def cos_loss(cosine, output1, output2):
    """reference: train_3d.py:86-92 (one scale per call, drawn from python's global `random`)."""
    index = random.randint(0, len(output1) - 1)
    sample1 = output1[index]
    sample2 = output2[index]
    if getattr(cosine, "returns_mean", False):
        c12, c21 = cosine(sample1[1], sample2[0].detach()), cosine(sample2[1], sample1[0].detach())
    else:
        c12, c21 = cosine(sample1[1], sample2[0].detach()).mean(), cosine(sample2[1], sample1[0].detach()).mean()
    loss = -(c12 + c21) * 0.5
    return loss, index


def seed_everything(seed):
    random.seed(seed)
    torch.manual_seed(seed)


def train_step(model, optimizer, batch, epoch, criterion, cosine, guard=True):
    """One iteration of train_3d.py:113-151.  Returns (loss, loss1, loss2, loss4, local_loss) as device
    scalars, or None if the divergence guard skipped the step."""
    input1, input2, gt, _gt2, local_views = batch
    bsz = input1.size(0)
    x1 = input1.float().cuda(non_blocking=True)
    x2 = input2.float().cuda(non_blocking=True)
    gt = gt.float().cuda(non_blocking=True)
    mask1, decoder_outputs1, middle_masks1 = model(x1)
    mask2, decoder_outputs2, _ = model(x2)
    loss2, index2 = cos_loss(cosine, decoder_outputs1, decoder_outputs2)
    local_loss = 0.0
    local_input = torch.cat([v.float().cuda(non_blocking=True) for v in local_views], dim=0)
    _, local_views_outputs, _ = model(local_input, local=True)
    local_views_outputs = [torch.stack(t) for t in local_views_outputs]
    for i in range(len(local_views)):
        local_views_outputs_tmp = [t[:, bsz * i: bsz * (i + 1)] for t in local_views_outputs]
        loss_local_1, _ = cos_loss(cosine, decoder_outputs1, local_views_outputs_tmp)
        loss_local_2, _ = cos_loss(cosine, decoder_outputs2, local_views_outputs_tmp)
        local_loss += loss_local_1
        local_loss += loss_local_2
    local_loss = local_loss / (2 * len(local_views))
    loss1 = criterion(mask1, gt)
    beta = 0.5 * (1. + math.cos(math.pi * epoch / 240))
    loss4 = beta * criterion(middle_masks1[index2], gt)
    loss = loss1 + loss2 + loss4 + local_loss
    if guard and epoch > 10 and loss > 1000:
        print('skip the step')
        return None
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach(), loss1.detach(), loss2.detach(), loss4.detach(), local_loss.detach()


def train_pcrlv2_3d(args, data_loader, out_channel=3):
    train_loader = data_loader['train']
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = 0
    if distributed:
        rank, _, local = _ddp.init_process_group_from_env()
        torch.cuda.set_device(local)
    seed_everything(getattr(args, "seed", 42))
    model = PCRLv23d()
    model = model.cuda()
    if getattr(args, "amp", False):
        model.set_compute_dtype(torch.bfloat16)
    optimizer = FusedSGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    dp = _ddp.DataParallel(model, optimizer) if distributed else None  # noqa: F841  (hooks into optimizer.step)

    criterion = MSELoss().cuda()
    cosine = CosineSimilarityMean().cuda()

    for epoch in range(0, args.epochs + 1):
        adjust_learning_rate(epoch, args, optimizer)
        if rank == 0:
            print("==> training...")
        time1 = time.time()
        loss, prob = train_pcrlv2_inner(args, epoch, train_loader, model, optimizer, criterion, cosine, verbose=(rank == 0))
        time2 = time.time()
        if rank == 0:
            print('epoch {}, total time {:.2f}'.format(epoch, time2 - time1))
        if rank == 0 and (epoch % 100 == 0 or epoch == 240):
            print('==> Saving...')
            state = {'opt': args, 'state_dict': model.state_dict(),
                     'optimizer': optimizer.state_dict(), 'epoch': epoch}
            save_file = os.path.join(args.output,
                                     args.model + "_" + args.n + '_' + args.phase + '_' + str(
                                         args.ratio) + '_' + str(epoch) + '.pt')
            torch.save(state, save_file)
            del state
        torch.cuda.empty_cache()
    return model


def train_pcrlv2_inner(args, epoch, train_loader, model, optimizer, criterion, cosine, verbose=True):
    """one epoch -- reference: train_3d.py:95-173"""
    model.train()
    batch_time = AverageMeter()
    data_time = AverageMeter()
    loss_meter = AverageMeter()
    mg_loss_meter = AverageMeter()
    prob_meter = AverageMeter()

    end = time.time()
    for idx, batch in enumerate(train_loader):
        data_time.update(time.time() - end)
        bsz = batch[0].size(0)
        out = train_step(model, optimizer, batch, epoch, criterion, cosine)
        if out is None:
            continue
        _, loss1, loss2, _, local_loss = out
        mg_loss_meter.update(loss1, bsz)
        loss_meter.update(loss2, bsz)
        prob_meter.update(local_loss, bsz)
        if (idx + 1) % 10 == 0:
            torch.cuda.synchronize()
        batch_time.update(time.time() - end)
        end = time.time()
        if verbose and (idx + 1) % 10 == 0:
            f = float
            print('Train: [{0}][{1}/{2}]\t'
                  'BT {3:.3f} ({4:.3f})\t'
                  'DT {5:.3f} ({6:.3f})\t'
                  'cos_loss {7:.3f} ({8:.3f})\t'
                  'mg loss {9:.3f} ({10:.3f})\t'
                  'local loss {11:.3f} ({12:.3f})'.format(
                      epoch, idx + 1, len(train_loader), batch_time.val, batch_time.avg, data_time.val, data_time.avg,
                      f(loss_meter.val), f(loss_meter.avg), f(mg_loss_meter.val), f(mg_loss_meter.avg),
                      f(prob_meter.val), f(prob_meter.avg)))
            sys.stdout.flush()
    return (float(mg_loss_meter.avg), float(prob_meter.avg))
