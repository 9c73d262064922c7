"""2D pre-training loop on the MI355X engine -- drop-in for the reference's train_2d.py  (SURVEY 8f N1).

Same entry point `train_pcrlv2(args, data_loader, out_channel=3)`, `cos_loss`, loss assembly (five scales, no divergence guard,
train_2d.py:139-171), LR schedule, log line and checkpoint (the ENCODER's state_dict only, train_2d.py:99).  The deliberate
differences are those listed in pcrlv2_amd/train_3d.py (bf16 for --amp, one process per GPU instead of nn.DataParallel, lazy
meters, --seed honoured).  `--encoder_weights FILE` initialises the encoder from a local torchvision-named ResNet-18 state_dict (the
reference downloads ImageNet weights at construction, pcrlv2_model.py:200; offline the default is random init, with a warning).
`--resume CKPT` continues from a checkpoint of the reference's 2D layout: that layout holds the ENCODER only (train_2d.py:99), so
the encoder, the epoch counter and -- when the shapes match -- the momentum buffers are restored, the decoder and heads restart.
"""
from __future__ import print_function

import math
import os
import random
import sys
import time

import torch

from . import config as _cfg
from . import ddp as _ddp
from . import functions as _fn
from . import ops as _ops
from .functions2d import MaskMSEFn, SegMSEFn, mse_loss2d
from .models.pcrlv2_model import PCRLv2
from .optim import FusedSGD
from .train_3d import BETA_PERIOD, COS_MAX_TERMS, CosineSimilarityMean, _fused_cos_losses, _to_gpu, cos_loss, seed_everything  # noqa: F401  (cos_loss: train_2d.py:111-117)
from .utils import AverageMeter, adjust_learning_rate


class MSELoss2d:
    """`criterion` of train_2d.py:78 on NHWC-memory predictions."""

    def cuda(self):
        return self

    def __call__(self, pred, target):
        return mse_loss2d(pred, target)


FUSED_STEP_2D = os.environ.get("PCRL_FUSED_STEP_2D", "1") != "0"     # A/B switch: 0 = the round-3 step (one launch per cosine mean, every map computed)


def step_losses(model, batch, epoch, criterion, cosine):
    """Forward half of one iteration (train_2d.py:139-168).  -> (total, restoration, global-cosine, deep-supervision, local-cosine)

    Engine form (our PCRLv2, our criterion / cosine objects): the 13 scale draws are taken from python's `random` FIRST -- the same 13
    `randint(0, 4)` calls in the same order as the reference's cos_loss calls (global pair; then for every local view (view 1, local_i),
    (view 2, local_i)); nothing in between consumes `random` -- so that the forwards know which deep-supervision map the step reads
    (masks1[scale of the first draw]) and skip the stateless work nobody reads (PCRLv2.forward_engine); all 26 cosine means in one launch,
    the local views concatenated in one launch, both restoration terms against the NCHW image without a layout copy, the total in one."""
    view1, view2, target, _unused_gt2, local_views = batch
    n = view1.size(0)
    target = _to_gpu(target)
    view1, view2 = _to_gpu(view1), _to_gpu(view2)
    nl = len(local_views)
    fused = (FUSED_STEP_2D and isinstance(model, PCRLv2) and isinstance(criterion, MSELoss2d) and getattr(cosine, "fusable", False)
             and 2 + 4 * nl <= COS_MAX_TERMS)
    _ops.fork_views(view1.device, path2d=True)     # config.VIEW_STREAMS_2D: the second view's forward (and backward) on its own stream
    if fused:
        ns = len(model.model.decoder.blocks)
        draws = [random.randint(0, ns - 1) for _ in range(1 + 2 * nl)]
        scale = draws[0]
        feats1, h1, low1 = model.forward_engine(view1, mask_scale=scale)
        with _ops.view_pass(view2.device, view2, path2d=True):
            feats2, _, _ = model.forward_engine(view2)
        loc = _ops.concat_batch([_to_gpu(v) for v in local_views])
        feats_loc, _, _ = model.forward_engine(loc)
        _ops.join_side_stream()                    # the cosine terms read both views' features on the main stream
        cos2, _ = _fused_cos_losses(feats1, feats2, feats_loc, n, nl, draws=draws)
        seg = model.model.segmentation_head[0]
        l_restore = SegMSEFn.apply(h1, seg.weight, seg.bias, target, model._seg)
        l_deep_raw = MaskMSEFn.apply(low1, target, 2 ** (ns - 1 - scale))
        beta = 0.5 * (1.0 + math.cos(math.pi * epoch / BETA_PERIOD))
        total, l_deep, l_global, l_local = _fn.loss_tail(l_restore, cos2, l_deep_raw, beta)
        return total, l_restore, l_global, l_deep, l_local
    feats1, mask1, masks1 = model(view1)
    with _ops.view_pass(view2.device, view2, path2d=True):
        feats2, _mask2, _ = model(view2)
    _ops.join_side_stream()                        # the cosine term below reads both views' features on the main stream
    l_global, scale = cos_loss(cosine, feats1, feats2)
    feats_loc, _, _ = model(torch.cat([_to_gpu(v) for v in local_views], dim=0), local=True)
    return assemble_losses(feats1, feats2, feats_loc, mask1, masks1, target, n, len(local_views), epoch, criterion, cosine, first=(l_global, scale))


def assemble_losses(feats1, feats2, feats_loc, mask1, masks1, target, n, nlocal, epoch, criterion, cosine, first=None):
    """train_2d.py:139-168 from the three forwards' outputs on: the global cosine term (its scale draw also picks the deep-supervision map), the
    2 * nlocal local terms in the reference's order -- (view 1, local_i), (view 2, local_i) for every local view i --, the restoration term,
    beta * the deep-supervision term, the sum.  Pure torch on whatever device the tensors live on: pinned on the CPU against the reference's own
    `cos_loss` and a restatement of its loop body (tests/golden/loss2d_*.npz, oracle/make_golden.py --loss2d).
    `first`: the (global term, scale) pair when the caller has already drawn it (step_losses draws it before the local views' forward, as the
    reference does).  -> (total, restoration, global-cosine, deep-supervision, local-cosine)"""
    l_global, scale = first if first is not None else cos_loss(cosine, feats1, feats2)
    stacked = [torch.stack(pair) for pair in feats_loc]                 # [2, 6n, C] per scale
    l_local = 0.0
    for i in range(nlocal):
        crop_i = [s[:, n * i: n * (i + 1)] for s in stacked]
        l_local = l_local + cos_loss(cosine, feats1, crop_i)[0]
        l_local = l_local + cos_loss(cosine, feats2, crop_i)[0]
    l_local = l_local / (2 * nlocal)
    l_restore = criterion(mask1, target)
    beta = 0.5 * (1.0 + math.cos(math.pi * epoch / BETA_PERIOD))
    l_deep = beta * criterion(masks1[scale], target)
    return l_restore + l_global + l_local + l_deep, l_restore, l_global, l_deep, l_local


def train_step(model, optimizer, batch, epoch, criterion, cosine):
    _ops.begin_step()
    _fn.reset_parked()
    dev = next(model.parameters()).device
    _ops.throttle_host(dev)      # at most config.MAX_STEPS_AHEAD steps of host run-ahead (allocator footprint, see config.py)
    losses = step_losses(model, batch, epoch, criterion, cosine)
    optimizer.zero_grad()
    losses[0].backward(gradient=_fn.root_gradient(losses[0]))
    optimizer.step()
    _ops.throttle_host(dev, step_done=True)
    # first complete step of this batch shape: size the allocator's per-stream pools for the steady state, once (ops.provision_allocator)
    _ops.provision_allocator(dev, key=("2d", tuple(batch[0].shape), len(batch[4])))
    return tuple(l.detach() for l in losses)


def train_pcrlv2(args, data_loader, out_channel=3):
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    # a group this call creates is this call's to take down (see train_3d.train_pcrlv2_3d)
    owns_group = distributed and not (torch.distributed.is_available() and torch.distributed.is_initialized())
    ok = False
    try:
        model = _train_pcrlv2(args, data_loader, distributed)
        ok = True
        return model
    finally:
        if owns_group:
            _ddp.shutdown(ok)


def _train_pcrlv2(args, data_loader, distributed):
    rank = 0
    if distributed:
        rank, _, local_rank = _ddp.init_process_group_from_env()
        torch.cuda.set_device(local_rank)
    seed_everything(getattr(args, "seed", 42))
    enc_w = getattr(args, "encoder_weights", None) or None
    chatty = rank == 0
    if enc_w is None and chatty:
        print("==> warning: encoder starts from RANDOM weights (no --encoder_weights); the reference starts from ImageNet ResNet-18")
    model = PCRLv2(encoder_weights=enc_w).cuda()
    if getattr(args, "amp", False):
        model.set_compute_dtype(torch.bfloat16)
    optimizer = FusedSGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    first_epoch = 0
    if getattr(args, "resume", None):
        # BEFORE the data-parallel wrapper is built: its initial broadcast then carries the resumed state from rank 0 to every rank
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.model.encoder.load_state_dict(ckpt["state_dict"])
        try:
            optimizer.load_state_dict(ckpt["optimizer"])
        except (ValueError, KeyError) as e:       # a checkpoint of another parameter list (group / size mismatch): keep fresh momentum, say so on EVERY rank
            print("==> [rank {}] optimizer state not restored: {}".format(rank, e))
        first_epoch = int(ckpt.get("epoch", -1)) + 1
    if distributed:
        _ddp.DataParallel(model, optimizer)
    criterion, cosine = MSELoss2d().cuda(), CosineSimilarityMean().cuda()
    if getattr(args, "resume", None):
        if chatty:
            print("==> resumed the ENCODER from {} (the 2D checkpoint layout holds nothing else); continuing with epoch {}".format(args.resume, first_epoch))
    for epoch in range(first_epoch, args.epochs + 1):
        adjust_learning_rate(epoch, args, optimizer)
        if chatty:
            print("==> training...")
        t_start = time.time()
        train_pcrlv2_inner(args, epoch, data_loader['train'], model, optimizer, criterion, cosine, verbose=chatty)
        if chatty:
            print('epoch {}, total time {:.2f}'.format(epoch, time.time() - t_start))
            if epoch % 100 == 0 or epoch == 240:     # train_2d.py:96-107: the ENCODER's weights only
                print('==> Saving...')
                model.flush_counters()
                state = {'opt': args, 'state_dict': model.model.encoder.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch}
                torch.save(state, os.path.join(args.output, "{}_{}_{}_{}_{}.pt".format(args.model, args.n, args.phase, args.ratio, epoch)))
        if _cfg.EMPTY_CACHE_PER_EPOCH:           # the reference's per-epoch empty_cache (train_3d.py:83 / train_2d.py:108); the steady-state pools are kept (ops.empty_cache)
            torch.cuda.empty_cache() if _cfg.EMPTY_CACHE_RAW else _ops.empty_cache()
    return model


def train_pcrlv2_inner(args, epoch, train_loader, model, optimizer, criterion, cosine, verbose=True):
    """One epoch (train_2d.py:120-195).  Returns (mean cosine loss, mean restoration loss, mean local loss)."""
    model.train()
    meters = {k: AverageMeter() for k in ("bt", "dt", "cos", "mg", "local")}
    tick = time.time()
    for it, batch in enumerate(train_loader, start=1):
        meters["dt"].update(time.time() - tick)
        out = train_step(model, optimizer, batch, epoch, criterion, cosine)
        n = batch[0].size(0)
        meters["mg"].update(out[1], n)
        meters["cos"].update(out[2], n)
        meters["local"].update(out[4], n)
        log_now = it % 10 == 0
        if log_now:
            torch.cuda.synchronize()
        meters["bt"].update(time.time() - tick)
        tick = time.time()
        if log_now and verbose:
            m = meters
            print('Train: [{0}][{1}/{2}]\t'
                  'BT {3:.3f} ({4:.3f})\t'
                  'DT {5:.3f} ({6:.3f})\t'
                  'cos_loss {7:.3f} ({8:.3f})\t'
                  'mg loss {9:.3f} ({10:.3f})\t'
                  'local loss {11:.3f} ({12:.3f})'.format(
                      epoch, it, len(train_loader), m["bt"].val, m["bt"].avg, m["dt"].val, m["dt"].avg,
                      float(m["cos"].val), float(m["cos"].avg), float(m["mg"].val), float(m["mg"].avg),
                      float(m["local"].val), float(m["local"].avg)))
            sys.stdout.flush()
    return float(meters["cos"].avg), float(meters["mg"].avg), float(meters["local"].avg)
