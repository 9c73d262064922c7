"""3D pre-training loop on the MI355X engine -- drop-in for the reference's train_3d.py.

Same entry point `train_pcrlv2_3d(args, data_loader, out_channel=3)`, same `cos_loss(cosine, output1, output2)`, same loss
assembly, LR schedule, log line, checkpoint dict and file name.  Differences, all deliberate:

  * compute: PCRLv23d, the losses and SGD run on hand-written gfx950 kernels (pcrlv2_amd.models / functions / optim);
  * `--amp` (apex O1 fp16 in the reference, train_3d.py:52-53) selects bfloat16 activations / MFMA operands with float32
    accumulation, statistics and master weights; bf16 needs no loss scaling;
  * `nn.DataParallel` (train_3d.py:54) -> one process per GPU + RCCL all-reduce (`pcrlv2_amd.ddp`): launch under torchrun,
    or plain `python main.py ...` for one GPU.  `--b` is the PER-PROCESS batch (the reference splits a global batch);
  * meters hold device scalars and are read only when the log line is printed: a step does not synchronise the GPU
    (the reference calls .item() twice and cuda.synchronize() every iteration);
  * the divergence guard (train_3d.py:140-142, `loss > 1000 and epoch > 10`) is decided ON THE DEVICE by default: a flag kernel,
    MAX-all-reduced under data parallelism, consumed by the SGD kernel (parameters and momentum stay bit-unchanged when it is set) --
    epochs 11..240 run without the forward -> backward host synchronisation the reference's `if loss > 1000` implies.  A skipped
    step still runs its backward (wasted work on a rare event) and its `.grad`s are this step's instead of the previous step's
    (nobody reads them: the next step starts with zero_grad); meters and the "skip the step" line are settled when the log line is
    printed.  PCRL_GUARD_SYNC=1 (or train_step(guard="sync")) restores the reference's host-side decision: the step returns None
    before zero_grad / backward / step;
  * `--seed`, ignored by the reference, seeds python `random` (the scale draws) and torch;
  * `--resume CKPT` (not in the reference, which only saves): restores model, momentum buffers and epoch from a checkpoint of
    the layout above -- written by this engine or by the reference -- and continues with the next epoch.
"""
from __future__ import print_function

import math
import os
import random
import sys
import time

import torch
import torch.distributed as dist

from . import config as _cfg
from . import ddp as _ddp
from . import functions as _fn
from . import ops as _ops
from .functions import cosine_mean, mse_loss
from .models import PCRLv23d
from .optim import FusedSGD
from .utils import AverageMeter, adjust_learning_rate

BETA_PERIOD = 240   # train_3d.py:136 hard-codes 240 (not args.epochs) in the deep-supervision weight; reproduced (Q2)


class MSELoss:
    """`criterion` of train_3d.py:56: mean squared error on the fused reduction kernel."""

    def cuda(self):
        return self

    def __call__(self, pred, target):
        return mse_loss(pred, target)


LAZY_SKIPS = True            # module attribute (tests / tools/step_ab.py flip it): the encoder stages' unpooled outputs are not stored by the step (PCRLv23d.forward lazy_skips)
SKIP_UNUSED_OUTPUTS = True   # module attribute, flipped by the bit-identity test (no environment switch since round 6)
COS_MAX_TERMS = 32   # csrc/heads_loss.hip
FUSED_COS_LOSSES = os.environ.get("PCRL_FUSED_COS", "1") != "0"   # all 26 cosine means of a step in one launch (0: one launch per mean)


class CosineSimilarityMean:
    """train_3d.py:57's nn.CosineSimilarity(), fused with the `.mean()` that every call site applies to it.  `fusable`: the training
    step may hand all of its cosine terms to one kernel launch instead of calling this object 26 times (same arithmetic)."""
    returns_mean = True
    fusable = True

    def cuda(self):
        return self

    def __call__(self, a, b):
        return cosine_mean(a, b)


def cos_loss(cosine, output1, output2):
    """Symmetric negative cosine similarity between predictor and (stop-gradient) projection on ONE scale drawn from
    python's global `random` -- train_3d.py:86-92.  output*[k] = [projection, prediction] of scale k."""
    k = random.randint(0, len(output1) - 1)
    (pro_a, pre_a), (pro_b, pre_b) = output1[k], output2[k]
    sim_ab, sim_ba = cosine(pre_a, pro_b.detach()), cosine(pre_b, pro_a.detach())
    if not getattr(cosine, "returns_mean", False):      # a plain nn.CosineSimilarity was passed in
        sim_ab, sim_ba = sim_ab.mean(), sim_ba.mean()
    return -(sim_ab + sim_ba) * 0.5, k


def seed_everything(seed):
    random.seed(seed)
    torch.manual_seed(seed)


def _to_gpu(t):
    return t.float().cuda(non_blocking=True)


def _fused_cos_losses(feats1, feats2, feats_loc, n, nlocal, draws=None):
    """The 13 cos_loss calls of train_3d.py:119-134 as ONE launch: the scales are drawn from python's `random` in the reference's
    order (global pair; then for every local view (view 1, local_i), (view 2, local_i)), the 26 cosine means and their weights
    (-1/2 per call; /(2 * nlocal) for the local group) go to pcrl_cosine_terms_*.  -> ([global term, local term] as one float32[2], first drawn
    scale): the pair goes to functions.loss_tail whole (selecting its elements here would cost select_backward's fills and adds)."""
    ns = len(feats1)
    tensors, idx = [], {}
    for name, fs in (("1", feats1), ("2", feats2), ("L", feats_loc)):
        for k in range(ns):
            idx[name, k, 0], idx[name, k, 1] = len(tensors), len(tensors) + 1          # 0 = projection, 1 = prediction
            tensors += [fs[k][0], fs[k][1]]
    spec = []

    def add(a, ra, b, rb, k, w, g):        # -(cos(pre_a, pro_b) + cos(pre_b, pro_a)) / 2 * w   at scale k
        spec.append((idx[a, k, 1], ra, idx[b, k, 0], rb, -0.5 * w, g))
        spec.append((idx[b, k, 1], rb, idx[a, k, 0], ra, -0.5 * w, g))

    # `draws`: the 1 + 2 * nlocal scales already taken from `random` in this order (train_2d.step_losses draws before the forwards)
    nxt = iter(draws).__next__ if draws is not None else (lambda: random.randint(0, ns - 1))
    k0 = nxt()
    add("1", 0, "2", 0, k0, 1.0, 0)
    wl = 1.0 / (2 * nlocal)
    for i in range(nlocal):
        add("1", 0, "L", n * i, nxt(), wl, 1)
        add("2", 0, "L", n * i, nxt(), wl, 1)
    return _fn.cosine_terms(spec, n, 2, tensors), k0


def step_losses(model, batch, epoch, criterion, cosine):
    """Forward half of one iteration (train_3d.py:113-138): three forwards, 13 cosine terms, two restoration terms.
    Returns (total, restoration, global-cosine, deep-supervision, local-cosine) as device scalars."""
    view1, view2, target, _unused_gt2, local_views = batch              # gt2 is never used by the reference either (Q3)
    n = view1.size(0)
    target = _to_gpu(target)
    # all cosine means of the step in one launch -- pcrl_cosine_terms_* takes up to 32 terms (2 + 4 per local view: up to 7 local views; the
    # reference's loop accepts any number): beyond that the step falls back to one launch per mean, same arithmetic
    fused = getattr(cosine, "fusable", False) and FUSED_COS_LOSSES and 2 + 4 * len(local_views) <= COS_MAX_TERMS
    view1, view2 = _to_gpu(view1), _to_gpu(view2)
    _ops.fork_views(view1.device)  # config.VIEW_STREAMS: the second view's forward (and backward) on its own stream, next to the first's
    _ops.prepack(model, view1.device)   # config.PREPACK: this step's packed / composed weight forms on the side stream, ahead of their use
    with _ops.deferred_join():     # the decoder stages' side branches (heads, deep-supervision maps) also run under the NEXT forward; joined on exit
        lz = dict(lazy_skips=True) if LAZY_SKIPS and isinstance(model, PCRLv23d) else {}      # the encoder stages' unpooled outputs have no reader in this step: not stored
        out1, feats1, masks1 = model(view1, **lz)
        # mask2, the local views' reconstruction and their deep-supervision maps are never used (train_3d.py:117,123): the engine's
        # model skips what has no state (out_tr, the trilinear upsampling) when asked for the features only; a plain nn.Module computes them
        fo = dict(features_only=True, **lz) if SKIP_UNUSED_OUTPUTS and isinstance(model, PCRLv23d) else dict(lz)
        with _ops.view_pass(view2.device, view2):
            _out2, feats2, _ = model(view2, **fo)
        if fused:
            loc = _ops.concat_batch([_to_gpu(v) for v in local_views])
            chunked = _ddp.chunk_partition_on()       # opt-in: nn.DataParallel's literal scatter of the [6B] local-view tensor (ddp.py)
            if chunked:
                loc = _ddp.chunk_local_inputs(loc, n, len(local_views))
            with _ops.view_pass(loc.device, loc, name="local"):
                _, feats_loc, _ = model(loc, local=True, **fo)
            if chunked:
                _ops.join_side_stream()        # the heads ran on the side stream (config.FWD_BRANCH_STREAM): the exchange reads their outputs on this one
                feats_loc = _ddp.chunk_local_features(feats_loc, n, len(local_views))
    if fused:
        cos2, scale = _fused_cos_losses(feats1, feats2, feats_loc, n, len(local_views))
        l_restore = criterion(out1, target)
        beta = 0.5 * (1.0 + math.cos(math.pi * epoch / BETA_PERIOD))
        total, l_deep, l_global, l_local = _fn.loss_tail(l_restore, cos2, criterion(masks1[scale], target), beta)   # one launch: the sum and beta * MSE
        return total, l_restore, l_global, l_deep, l_local
    l_global, scale = cos_loss(cosine, feats1, feats2)
    loc = torch.cat([_to_gpu(v) for v in local_views], dim=0)
    if _ddp.chunk_partition_on():
        loc = _ddp.chunk_local_inputs(loc, n, len(local_views))
    _, feats_loc, _ = model(loc, local=True, **fo)
    if _ddp.chunk_partition_on():
        feats_loc = _ddp.chunk_local_features(feats_loc, n, len(local_views))
    stacked = [torch.stack(pair) for pair in feats_loc]                 # [2, 6n, C] per scale
    l_local = 0.0
    for i in range(len(local_views)):
        crop_i = [s[:, n * i: n * (i + 1)] for s in stacked]
        l_local = l_local + cos_loss(cosine, feats1, crop_i)[0]
        l_local = l_local + cos_loss(cosine, feats2, crop_i)[0]
    l_local = l_local / (2 * len(local_views))
    l_restore = criterion(out1, target)
    beta = 0.5 * (1.0 + math.cos(math.pi * epoch / BETA_PERIOD))
    l_deep = beta * criterion(masks1[scale], target)                    # the scale drawn by the FIRST cos_loss call
    return l_restore + l_global + l_deep + l_local, l_restore, l_global, l_deep, l_local


def begin_step():
    """Reset the per-step engine state (ops pass counter; gradients parked by a backward that raised)."""
    _ops.begin_step()
    _fn.reset_parked()


GUARD_THRESHOLD = 1000.0   # train_3d.py:140
GUARD_FIRST_EPOCH = 11     # `epoch > 10`
GUARD_SYNC = os.environ.get("PCRL_GUARD_SYNC", "0") == "1"


class StepLosses(tuple):
    """(loss, loss1, loss2, loss4, local_loss) as detached device scalars, plus `.skipped`: None, or the device flag (float32[1]) of the
    divergence guard -- 1 when the update of this step was skipped on the device."""
    skipped = None


def divergence_flag(loss, group=None, collective=True):
    """float32[1] on loss's device: 1 if loss > 1000 on ANY rank of `group` (the reference is one process with one loss and one decision;
    with one process per GPU the decision must be collective -- a rank that skipped alone would leave its peers waiting in the gradient
    all-reduce).  `collective=False`: this process's own decision (no data-parallel wrapper: nobody to agree with)."""
    flag = torch.empty(1, dtype=torch.float32, device=loss.device)
    if loss.is_cuda:
        _ops.lib().call("pcrl_guard_flag", loss.detach().float().reshape(1), GUARD_THRESHOLD, flag, _ops.stream_handle())
    else:       # host tensors (the gloo tests of the collective form): same predicate
        flag[0] = 1.0 if float(loss) > GUARD_THRESHOLD else 0.0
    if collective and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag


def train_step(model, optimizer, batch, epoch, criterion, cosine, guard=True):
    """One iteration of train_3d.py:113-151.  Returns StepLosses (loss, loss1, loss2, loss4, local_loss: detached device scalars).
    Divergence guard (epoch > 10, loss > 1000): by default decided on the device -- the returned `.skipped` flag says whether the SGD
    kernel left the parameters alone; with guard="sync" (PCRL_GUARD_SYNC=1) on the host like the reference -- None is returned before
    zero_grad / backward / step.  guard=False: no guard."""
    begin_step()            # forward-pass numbering / parked-gradient state start clean even after a skipped or failed step
    dev = next(model.parameters()).device
    _ops.throttle_host(dev)      # at most config.MAX_STEPS_AHEAD steps of run-ahead (allocator footprint, see config.py)
    with _ops.trace_range("forward"):
        losses = step_losses(model, batch, epoch, criterion, cosine)
    flag = None
    if guard and epoch >= GUARD_FIRST_EPOCH:
        # the decision is taken by the ranks that share gradients: the data-parallel wrapper's group (a wrapper on a sub-group must not pull
        # ranks outside it into this collective); without a wrapper the process decides alone
        dp = getattr(optimizer, "data_parallel", None)
        flag = divergence_flag(losses[0], group=getattr(dp, "group", None), collective=dp is not None and getattr(dp, "_active", False))
        if guard == "sync" or GUARD_SYNC:
            if bool(flag):       # device -> host read: the reference's semantics, and its synchronisation
                print('skip the step')
                return None
            flag = None
    optimizer.zero_grad()
    with _ops.trace_range("backward"):
        losses[0].backward(gradient=_fn.root_gradient(losses[0]))
    optimizer.skip_flag = flag
    with _ops.trace_range("optimizer"):
        optimizer.step()
    _ops.throttle_host(dev, step_done=True)
    # first complete step of this batch shape: size the allocator's per-stream pools for the steady state, once (ops.provision_allocator)
    _ops.provision_allocator(dev, key=("3d", tuple(batch[0].shape), len(batch[4])))
    out = StepLosses(l.detach() for l in losses)
    out.skipped = flag
    return out


def _checkpoint_name(args, epoch):
    return os.path.join(args.output, "{}_{}_{}_{}_{}.pt".format(args.model, args.n, args.phase, args.ratio, epoch))


def load_checkpoint(path, model, optimizer=None):
    """Load a checkpoint of the layout train_3d.py:71-82 writes ({'opt','state_dict','optimizer','epoch'}); keys saved from a
    DataParallel-wrapped model ('module.' prefix, train_3d.py:54) are accepted.  Returns the stored epoch."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in ckpt["state_dict"].items()}
    model.load_state_dict(sd)
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return int(ckpt.get("epoch", -1))


def train_pcrlv2_3d(args, data_loader, out_channel=3):
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    # a group this call creates is this call's to take down (ddp.shutdown: barrier + destroy_process_group, also when an exception propagates):
    # nn.DataParallel (train_3d.py:54) needs no teardown, one process per GPU does -- ranks that return with the group alive abort now and then
    owns_group = distributed and not (torch.distributed.is_available() and torch.distributed.is_initialized())
    ok = False
    try:
        model = _train_pcrlv2_3d(args, data_loader, distributed)
        ok = True
        return model
    finally:
        if owns_group:
            _ddp.shutdown(ok)


def _train_pcrlv2_3d(args, data_loader, distributed):
    rank = 0
    if distributed:
        rank, _, local_rank = _ddp.init_process_group_from_env()
        torch.cuda.set_device(local_rank)
    seed_everything(getattr(args, "seed", 42))
    model = PCRLv23d().cuda()
    if getattr(args, "amp", False):
        model.set_compute_dtype(torch.bfloat16)
    optimizer = FusedSGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    chatty = rank == 0
    first_epoch = 0
    if getattr(args, "resume", None):       # before the data-parallel wrapper: its initial broadcast carries rank 0's resumed state to every rank
        first_epoch = load_checkpoint(args.resume, model, optimizer) + 1
        if chatty:
            print("==> resumed from {} (continuing with epoch {})".format(args.resume, first_epoch))
    if distributed:
        _ddp.DataParallel(model, optimizer)          # hooks itself into optimizer.step()
    criterion, cosine = MSELoss().cuda(), CosineSimilarityMean().cuda()

    for epoch in range(first_epoch, args.epochs + 1):          # inclusive upper bound, like the reference (Q1): lr reaches 0 in the last epoch
        adjust_learning_rate(epoch, args, optimizer)
        if chatty:
            print("==> training...")
        t_start = time.time()
        train_pcrlv2_inner(args, epoch, data_loader['train'], model, optimizer, criterion, cosine, verbose=chatty)
        if chatty:
            print('epoch {}, total time {:.2f}'.format(epoch, time.time() - t_start))
            if epoch % 100 == 0 or epoch == 240:     # checkpoint cadence and layout of train_3d.py:71-82
                print('==> Saving...')
                torch.save({'opt': args, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch},
                           _checkpoint_name(args, epoch))
        if _cfg.EMPTY_CACHE_PER_EPOCH:           # the reference's per-epoch empty_cache (train_3d.py:83 / train_2d.py:108); the steady-state pools are kept (ops.empty_cache)
            torch.cuda.empty_cache() if _cfg.EMPTY_CACHE_RAW else _ops.empty_cache()
    return model


def train_pcrlv2_inner(args, epoch, train_loader, model, optimizer, criterion, cosine, verbose=True):
    """One epoch (train_3d.py:95-173).  Returns (mean restoration loss, mean local loss)."""
    model.train()
    meters = {k: AverageMeter() for k in ("bt", "dt", "cos", "mg", "local")}
    skipped_flags = []
    tick = time.time()
    for it, batch in enumerate(train_loader, start=1):
        meters["dt"].update(time.time() - tick)
        out = train_step(model, optimizer, batch, epoch, criterion, cosine)
        if out is None:         # host-side guard (PCRL_GUARD_SYNC=1): the reference's `continue`
            continue
        n = batch[0].size(0)
        vals = (out[1], out[2], out[4])
        if out.skipped is not None:
            # guard decided on the device: a skipped step must not enter the meters (the reference `continue`s before them) -- its weight is
            # n * (1 - skipped), a device scalar, AND its values are masked to 0 (a diverged loss is often inf / NaN: inf * 0 would poison the
            # running sums); the "skip the step" lines are printed when the flags are read, with the log line
            live = 1.0 - out.skipped.reshape(())
            n = n * live
            vals = tuple(torch.where(live > 0, v, torch.zeros_like(v)) for v in vals)
            skipped_flags.append(out.skipped)
        meters["mg"].update(vals[0], n)
        meters["cos"].update(vals[1], n)
        meters["local"].update(vals[2], n)
        log_now = it % 10 == 0
        if log_now:
            torch.cuda.synchronize()
            if skipped_flags and verbose:
                for _ in range(int(torch.cat(skipped_flags).sum().item())):
                    print('skip the step')
            skipped_flags.clear()
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
    if skipped_flags and verbose:       # steps skipped after the last log line of the epoch
        for _ in range(int(torch.cat(skipped_flags).sum().item())):
            print('skip the step')
    return float(meters["mg"].avg), float(meters["local"].avg)
