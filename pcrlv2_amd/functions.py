"""autograd.Function wrappers: the drop-in boundary towards torch's autograd engine.

One Function per reference stage so every gradient fan-out inside a stage (UpTransition feeds the
next stage, the pooled projection head and the deep-supervision head from the same tensor) is
combined by our own kernels instead of autograd's adds.  Unused outputs arrive as None
(`set_materialize_grads(False)`) and whole branches are skipped, which reproduces what the
reference's autograd does for the never-used `mask2` / view-2 deep-supervision heads.
"""
from __future__ import annotations

import weakref

import torch
from torch.autograd import Function

from . import config, ops
from ._lib import ACT_RELU, ACT_SIGMOID, ACT_SILU


def _act_grad(g, dtype):
    return ops.to_act(g, dtype)


# ---- parameter gradients: delivered to `.grad` by the engine, not by autograd's AccumulateGrad -------------------------------
# A step runs three forwards over the same parameters (view 1, view 2, local views; train_3d.py:118-121), so autograd would
# launch one tiny `grad += g` kernel per parameter and extra pass (218 launches per step).  Instead the stage Functions return
# None for their parameters and park the gradient tensors here; when the backward pass ends (engine callback) -- or earlier,
# bucket by bucket, when the data-parallel wrapper asks -- all parked gradients are summed with three multi-tensor launches,
# in autograd's own order (g_first + g_second + g_third), straight into the optimizer's flat gradient arena when the parameter
# has a slot there (`p._pcrl_gview`, set by FusedSGD), and `.grad` is set like AccumulateGrad would have set it.
# `torch.autograd.grad(loss, params)` does not accumulate and therefore needs config.DIRECT_PARAM_GRADS = False.
_parked = {}            # id(parameter) -> (parameter, [gradient tensors in arrival order])
_parked_events = {}     # id(parameter) -> [event recorded on the producing stream behind each stage's gradients] (data-parallel overlap only)
_callback_queued = False
_final_callback = None  # set by ddp.DataParallel: callable(param), a parameter's gradient is complete for this step
_finalize_hook = None   # set by ddp.DataParallel: callable() replacing the default end-of-backward flush


def set_ddp_callbacks(final_callback, finalize_hook):
    global _final_callback, _finalize_hook
    _final_callback, _finalize_hook = final_callback, finalize_hook


def _deliver_composed(only=None):
    """Composed up-conv stages (ops.ComposedUpConv): the chain rule from the accumulated gradient of the composed weights to
    up_conv.weight / up_conv.bias / ops.0.conv1.weight, once per backward() call."""
    ev = None
    for p, g in ops.deliver_composed(only):
        if p.requires_grad:
            _parked.setdefault(id(p), (p, []))[1].append(g)
            if _final_callback is not None:
                if ev is None:      # one event behind the chain rule, on the stream it ran on (ops.deliver_composed: the side stream when it is active)
                    st = ops.side_stream(g.device) if (config.EARLY_COMPOSED and ops.side_wgrad(g.device).active) else torch.cuda.current_stream(g.device)
                    ev = torch.cuda.Event()
                    ev.record(st)
                _parked_events.setdefault(id(p), []).append(ev)
                _final_callback(p)


def _end_of_backward():
    global _callback_queued
    _callback_queued = False
    _deliver_composed()
    if _finalize_hook is not None:
        _finalize_hook()
    else:
        flush_param_grads()


def _queue_end_of_backward():
    global _callback_queued
    if not _callback_queued:
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        _callback_queued = True


def _park(p, g):
    """Take the gradient `g` of parameter `p` out of autograd's hands (returns what the Function hands to autograd instead)."""
    global _callback_queued
    if g is None:
        return g
    if not config.DIRECT_PARAM_GRADS:
        ops.join_side_stream()     # autograd's AccumulateGrad reads g on the main stream
        return g
    if not p.requires_grad:
        return None
    _parked.setdefault(id(p), (p, []))[1].append(g)
    _queue_end_of_backward()
    return None


def reset_parked():
    """Start of a step: drop gradients parked by a backward pass that raised before its end-of-backward callback ran (the engine
    discards the queued callback then, and a stale flag would park every later gradient without ever delivering it)."""
    global _callback_queued
    _parked.clear()
    _parked_events.clear()
    _callback_queued = False


def parked_params():
    return [p for p, _ in _parked.values()]


def take_parked(p):
    _parked_events.pop(id(p), None)
    return _parked.pop(id(p), (p, []))[1]


def _flush_fused(items):
    """The common case in one launch (pcrl_grad_sum; config.FUSED_GRAD_SUM): every parameter lives in ONE FusedSGD arena, has no gradient yet
    (no accumulation across backward() calls) and at most 8 contiguous float32 terms.  -> False: the caller's generic
    path (multi-tensor copies and adds) runs instead.  Same additions in the same order: bit-identical."""
    if not config.FUSED_GRAD_SUM:
        return False
    opt, idx = None, {}
    nsrc = 1
    for p, gs in items:
        slot = getattr(p, "_pcrl_gslot", None)
        if slot is None or p.grad is not None or (opt is not None and slot[0] is not opt) or len(gs) > 8:
            return False
        for g in gs:
            if not ops.is_shared_zero(g) and (g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != p.numel() or g.data_ptr() % 4
                                              or g.device != p.device):
                return False
        opt = slot[0]
        idx[slot[1]] = (p, gs)
        nsrc = max(nsrc, len(gs))
    import ctypes
    t0, t1 = min(idx), max(idx) + 1
    srcs = (ctypes.c_void_p * ((t1 - t0) * nsrc))()
    for i, (p, gs) in idx.items():
        # the first term is copied even when it is the shared zero vector (a bias in front of a BatchNorm: its gradient IS zero), later zero terms add nothing
        for k, g in enumerate(gs):
            if k == 0 or not ops.is_shared_zero(g):
                srcs[(i - t0) * nsrc + k] = g.data_ptr()
        p.grad = p._pcrl_gview
    ops.lib().call("pcrl_grad_sum", opt.flat_g, opt._offsets, opt._numels, ctypes.addressof(opt._offsets_c), ctypes.addressof(srcs), t0, t1 - t0, nsrc,
                   ops.stream_handle())
    return True


@torch.no_grad()
def flush_param_grads(params=None, on_stream=None):
    """Sum the parked gradients of `params` (default: all) into `.grad`, level by level with multi-tensor launches.
    on_stream: run the sums THERE (the data-parallel wrapper's communication stream, for a bucket that is final mid-backward): that stream
    waits for the producers (current, side and view streams) and the main stream is not joined with anything -- the weight gradients still
    queued on the side stream keep running next to the data-gradient chain instead of being waited for at every bucket."""
    keys = list(_parked.keys()) if params is None else [id(p) for p in params if id(p) in _parked]
    items = [_parked.pop(k) for k in keys]
    if not items:
        return
    import contextlib
    ctx = contextlib.nullcontext()
    if on_stream is None:
        ops.join_side_stream()        # weight gradients are produced on a side stream (ops.side_wgrad)
    else:
        # wait for exactly the kernels that produced these gradients: the event each stage's backward left behind them on its stream
        # (mark_final).  Waiting for the producer STREAMS instead would make a bucket that is final early in view 1's backward wait for the
        # whole backward of view 2 (all of it is queued on the view stream by then) -- and everything queued behind the bucket on this
        # stream, the remaining weight gradients, with it: measured +1.1 ms per step on a one-rank group.
        dev = items[0][0].device
        seen, missing = set(), False
        for k in keys:
            evs = _parked_events.pop(k, None)
            if evs is None:
                missing = True
                continue
            for ev in evs:
                if id(ev) not in seen:
                    seen.add(id(ev))
                    on_stream.wait_event(ev)
        side = ops.producer_streams(dev)[:1]          # weight gradients: the side stream, in launch order
        if side and side[0].cuda_stream != on_stream.cuda_stream:
            on_stream.wait_stream(side[0])
        if missing:                                    # a gradient parked outside a stage backward: fall back to the streams
            on_stream.wait_stream(torch.cuda.current_stream(dev))
            for st in ops.producer_streams(dev):
                if st.cuda_stream != on_stream.cuda_stream:
                    on_stream.wait_stream(st)
        for _, gs in items:
            for g in gs:
                g.record_stream(on_stream)      # allocated on a producer stream, read here: keep the allocator from recycling it under us
        ctx = torch.cuda.stream(on_stream)
    with ctx:
        if _flush_fused(items):
            return
        targets, level = [], 0
        copy_dst, copy_src = [], []
        for p, gs in items:
            if p.grad is not None:
                targets.append((p.grad, gs, 0))            # accumulation across backward() calls: add everything
                continue
            view = getattr(p, "_pcrl_gview", None)
            if view is not None:
                copy_dst.append(view)
                copy_src.append(gs[0])
                p.grad = view
            else:
                own = gs[0].is_contiguous() and gs[0].shape == p.shape and not ops.is_shared_zero(gs[0])
                p.grad = gs[0] if own else gs[0].reshape(p.shape).clone()
            targets.append((p.grad, gs, 1))
        if copy_dst:
            torch._foreach_copy_(copy_dst, copy_src)
        while True:
            pairs = [(t, gs[first + level]) for t, gs, first in targets if first + level < len(gs)]
            if not pairs:
                break
            pairs = [(t, g) for t, g in pairs if not ops.is_shared_zero(g)]     # conv biases in front of a BatchNorm: exactly zero
            if pairs:
                torch._foreach_add_([t for t, _ in pairs], [g for _, g in pairs])
            level += 1


def mark_final(ctx, params):
    """End of a stage's backward.  Under the data-parallel wrapper: leave ONE event on the current stream behind the gradients this stage
    just parked (flush_param_grads(on_stream=...) waits for it), and -- if the stage ran in the FIRST forward of the step: nothing will add
    to these gradients any more -- report the parameters final."""
    if _final_callback is None:
        return
    ev = None
    for p in params:
        if id(p) in _parked and p.is_cuda:       # (the CPU tests of the wrapper's bookkeeping have no streams)
            if ev is None:
                ev = torch.cuda.Event()
                ev.record()
            _parked_events.setdefault(id(p), []).append(ev)
    if getattr(ctx, "pass_idx", 1) == 0:
        for p in params:
            _final_callback(p)


def _slope(mod):
    """nn.PReLU's slope vector of a LUConv built with act='prelu' (constructor variant), else None."""
    return mod.activation.weight if getattr(mod, "_prelu", False) else None


def _park_slope(mod, sv):
    """The PReLU slope's gradient does not travel through autograd (the slope is not an input of the stage Functions): parked like every
    other parameter gradient.  Needs the engine-delivered gradients (config.DIRECT_PARAM_GRADS)."""
    if sv.prelu is None or sv.dslope is None:
        return
    if not config.DIRECT_PARAM_GRADS:
        raise RuntimeError("act='prelu' needs config.DIRECT_PARAM_GRADS (the slope gradient is delivered by the engine, not by autograd)")
    _park(mod.activation.weight, sv.dslope)
    sv.dslope = None


def _remember_output(mod, sv, a):
    """ops.0 of an nn.Sequential(LUConv, LUConv) (models/pcrlv2_model_3d.py:37-45) remembers -- weakly -- what it just produced, so that ops.1's
    forward, which runs next, can recognise its input as that activation (LUConv._below, _below_saved)."""
    mod._last_out = (weakref.ref(sv), a.data_ptr(), tuple(a.shape))


def _below_saved(mod, x):
    """-> the saved state of the LUConv whose activation `x` is, if `mod` is the ops.1 of a pair and x is exactly what ops.0 returned in this
    pass; else None.  That activation then has this convolution as its only consumer by construction of the model, and backward double-checks:
    the fused first pass is used only if the gradient that reaches ops.0 is the very tensor this layer's data gradient wrote (ops.take_pre_partial)."""
    below = getattr(mod, "_below", None)
    last = getattr(below, "_last_out", None) if below is not None else None
    if last is None or not config.DGRAD_BNRED:
        return None
    sv = last[0]()
    if sv is None or x.data_ptr() != last[1] or tuple(x.shape) != last[2]:
        return None
    return sv


class LUConvFn(Function):
    """act(bn1(conv1(x)))  --  models/pcrlv2_model_3d.py:32-34."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, mod):
        dt = mod.compute_dtype
        gn = getattr(mod, "_gn_groups", 0)
        w_eff, ctx.ci = w, 0
        if getattr(mod, "_ci_pad", 0):         # in_channels != 1 (constructor variant): zero-padded to the implicit-GEMM kernels' channel granule
            ctx.ci = w.shape[1]
            x, w_eff = ops.pad_first_layer(x, w, mod._ci_pad, dt)
        a, sv = ops.luconv_forward(x, w_eff, b, gamma, beta, None if gn else mod.bn1.running_mean, None if gn else mod.bn1.running_var,
                                   mod._packed, mod._act, dt, gn_groups=gn, prelu=_slope(mod), inorm=getattr(mod, "_inorm", False))
        mod._count_batch()
        ctx.sv, ctx.mod, ctx.dt = sv, mod, dt
        ctx.below = _below_saved(mod, x)
        _remember_output(mod, sv, a)
        ctx.wref, ctx.gref = w_eff, gamma
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.plist = (w, b, gamma, beta)
        ctx.set_materialize_grads(False)
        return a

    @staticmethod
    def backward(ctx, da):
        if da is None:
            return (None,) * 6
        sv = ctx.sv
        da = da.contiguous() if sv.kind == "to1" else _act_grad(da, ctx.dt)
        dx, dw, db, dg, dbeta = ops.luconv_backward(sv, da, ctx.wref, ctx.gref, ctx.mod._packed, ctx.dt,
                                                    need_dx=ctx.needs_input_grad[0] and sv.kind != "c1", bnred=ctx.below)
        if ctx.ci:                              # padded first layer: the real input channels' share
            ops.join_side_stream()              # the weight gradient was produced on the side stream; the slice below runs on this one
            dw = dw[:, :ctx.ci].contiguous()
            dx = None if dx is None else dx[:, :ctx.ci]
        _park_slope(ctx.mod, sv)
        w, b, gamma, beta = ctx.plist
        out = dx, _park(w, dw), _park(b, db), _park(gamma, dg), _park(beta, dbeta), None
        mark_final(ctx, ctx.plist)
        return out


class LUConvPoolFn(Function):
    """(a, p) = (act(bn1(conv1(x))), MaxPool3d(2)(a))  --  the second LUConv of an encoder stage and the `self.maxpool` that follows it
    (models/pcrlv2_model_3d.py:32-34,115-117) as ONE autograd node.  When only the pooled tensor carries a gradient (the skip tensors
    are never consumed) max_pool3d_backward is folded into the BatchNorm backward passes (ops.bn_act_backward, pool_dp):
    the full-resolution gradient of `a` is never written or read."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, mod, pool_only=False):
        dt = mod.compute_dtype
        (a, p), sv = ops.luconv_forward(x, w, b, gamma, beta, mod.bn1.running_mean, mod.bn1.running_var, mod._packed, mod._act, dt, pooled=True,
                                        prelu=_slope(mod), pool_only=pool_only)
        mod._count_batch()
        ctx.sv, ctx.mod, ctx.dt = sv, mod, dt
        ctx.below = _below_saved(mod, x)
        ctx.wref, ctx.gref = w, gamma
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.plist = (w, b, gamma, beta)
        ctx.set_materialize_grads(False)
        ctx.pool_only = a is None
        mod._last_saved = sv            # for PCRLv23d's lazily materialised skip attribute (pool_only: `a` was not stored)
        return p if a is None else (a, p)

    @staticmethod
    def backward(ctx, *grads):
        da, dp = (None, grads[0]) if ctx.pool_only else grads
        if da is None and dp is None:
            return (None,) * 7
        sv, dt = ctx.sv, ctx.dt
        N, D, H, W, _, Co = sv.geom
        pool_dp = None
        if dp is not None:
            dp = _act_grad(dp, dt)
            if da is None and sv.prelu is None and ops.bn_pool_ok(D, H, W, Co, dt):
                pool_dp = dp
            else:   # somebody consumed the full-resolution activation too: materialise max_pool3d_backward and add
                a = ops.prelu_forward(sv.z, sv.prelu, dt) if sv.prelu is not None else ops.bn_act_apply(sv.y, sv.scale, sv.shift, N * D * H * W, Co, sv.act, dt)
                dfull = ops.maxpool_backward(a, dp, dt)
                da = dfull if da is None else _act_grad(da, dt) + dfull
        elif da is not None:
            da = _act_grad(da, dt)
        dx, dw, db, dg, dbeta = ops.luconv_backward(sv, da, ctx.wref, ctx.gref, ctx.mod._packed, dt, need_dx=ctx.needs_input_grad[0],
                                                    pool_dp=pool_dp, bnred=ctx.below)
        _park_slope(ctx.mod, sv)
        w, b, gamma, beta = ctx.plist
        out = dx, _park(w, dw), _park(b, db), _park(gamma, dg), _park(beta, dbeta), None, None
        mark_final(ctx, ctx.plist)
        return out


class MaxPoolFn(Function):
    """nn.MaxPool3d(2)  --  models/pcrlv2_model_3d.py:100,115-117."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.x, ctx.dt = x, dt
        ctx.set_materialize_grads(False)
        return ops.maxpool_forward(x, dt)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        return ops.maxpool_backward(ctx.x, _act_grad(dy, ctx.dt), ctx.dt), None


class UpStageFn(Function):
    """UpTransition.forward  --  models/pcrlv2_model_3d.py:62-72.

    inputs : x, up_w, up_b, [w,b,gamma,beta] of ops.0, ops.1, bn.(gamma,beta), ph0.(w,b), ph1.(gamma,beta),
             ph3.(w,b), [w,b,gamma,beta] of deep_supervision_head, module
    outputs: x_out (activation), x_pro [N,C], x_pre [N,C], x_mask [N,1,D,H,W] float32
    """

    @staticmethod
    def forward(ctx, x, up_w, up_b, w0, b0, g0, be0, w1, b1, g1, be1, bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b,
                dw_, db_, dg_, dbe_, mod):
        dt = mod.compute_dtype
        x = ops.to_act(x, dt)
        l0, l1, ld = mod.ops[0], mod.ops[1], mod.deep_supervision_head
        gn = getattr(l0, "_gn_groups", 0)
        rs = lambda m: (None, None) if gn else (m.bn1.running_mean, m.bn1.running_var)
        s0, s1 = _slope(l0), _slope(l1)
        if config.COMPOSE_UPCONV and not gn and s0 is None:
            # up_conv and ops.0's conv1 are two linear maps with nothing in between (:64): one 8-tap operator on the coarse grid,
            # the upsampled tensor is never formed (ops.upconv_luconv_forward, csrc/upconv_fused.hip)
            a0, sv0 = ops.upconv_luconv_forward(x, up_w, up_b, w0, b0, g0, be0, *rs(l0), mod._composed_up, l0._act, dt)
        else:
            up = ops.convt_forward(x, up_w, up_b, mod._packed_up, dt)
            a0, sv0 = ops.luconv_forward(up, w0, b0, g0, be0, *rs(l0), l0._packed, l0._act, dt, gn_groups=gn, prelu=s0)
        if gn:
            a1, sv1 = ops.luconv_forward(a0, w1, b1, g1, be1, *rs(l1), l1._packed, l1._act, dt, gn_groups=gn, prelu=s1)
            g = ops.gap_forward(a1, dt)
        else:   # activation and its global average pool (:67) from one pass over the convolution output
            (a1, g), sv1 = ops.luconv_forward(a0, w1, b1, g1, be1, *rs(l1), l1._packed, l1._act, dt, gap=True, prelu=s1)
        # The stage's side branches -- nothing in the forward consumes them -- on the side stream, next to the next stage's convolutions
        # (config.FWD_BRANCH_STREAM; the main stream joins at the end of the forward).  Backward runs on the main stream as before.
        with ops.side_branch(a1.device, a1, g):
            x_pro, m_pro, r_pro = ops.bn1d_forward(g, bn_g, bn_b, mod.bn.running_mean, mod.bn.running_var, relu=False)
            h0 = ops.linear_forward(x_pro, p0_w, p0_b)
            ph1 = mod.predictor_head[1]
            h1, m_h, r_h = ops.bn1d_forward(h0, p1_g, p1_b, ph1.running_mean, ph1.running_var, relu=True)
            x_pre = ops.linear_forward(h1, p3_w, p3_b)
            x_mask, svd = ops.luconv_forward(a1, dw_, db_, dg_, dbe_, ld.bn1.running_mean, ld.bn1.running_var, ld._packed, ACT_SIGMOID, dt,
                                             inorm=getattr(ld, "_inorm", False))
        for m in (l0, l1, ld):
            m._count_batch()
        mod._count_batch_heads()
        ctx.mod, ctx.dt = mod, dt
        svd.x = None   # = a1, an OUTPUT of this Function: re-attached from saved_tensors in backward (see below)
        ctx.x, ctx.sv0, ctx.sv1, ctx.svd = x, sv0, sv1, svd
        # OUTPUTS needed in backward go through save_for_backward: stashing an output on ctx directly makes a reference
        # cycle (output -> grad_fn -> ctx -> output) that only the cyclic GC frees -- tens of GB of activations linger.
        ctx.save_for_backward(a1, x_pro)
        ctx.heads = (g, m_pro, r_pro, h0, h1, m_h, r_h)   # intermediates (no grad_fn): safe on ctx
        ctx.params = (up_w, w0, g0, w1, g1, bn_g, p0_w, p1_g, p3_w, dw_, dg_)
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.plist = (up_w, up_b, w0, b0, g0, be0, w1, b1, g1, be1, bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b, dw_, db_, dg_, dbe_)
        ctx.set_materialize_grads(False)
        return a1, x_pro, x_pre, x_mask

    @staticmethod
    def backward(ctx, d_out, d_pro, d_pre, d_mask):
        n_in = 24
        if d_out is None and d_pro is None and d_pre is None and d_mask is None:
            return (None,) * n_in
        mod, dt = ctx.mod, ctx.dt
        up_w, w0, g0, w1, g1, bn_g, p0_w, p1_g, p3_w, dsw, dsg = ctx.params
        a1, x_pro = ctx.saved_tensors
        ctx.svd.x = a1
        g, m_pro, r_pro, h0, h1, m_h, r_h = ctx.heads
        l0, l1, ld = mod.ops[0], mod.ops[1], mod.deep_supervision_head
        grads = [None] * n_in
        row_g = None

        # ---- gradient w.r.t. a1 (output of ops.1): up to three sources, combined by our kernels ----
        d_a1 = _act_grad(d_out, dt) if d_out is not None else None
        # predictor / projection heads (train: pcrlv2_model_3d.py:55-59,69-70)
        if d_pre is not None or d_pro is not None:
            # ctx.plist[10:18] = bn.(weight, bias), predictor_head.0.(weight, bias), .1.(weight, bias), .3.(weight, bias)
            d_g, hg = ops.heads_backward(d_pro, d_pre, ctx.heads, x_pro, ctx.plist[10:18])
            grads[11], grads[12] = hg[0], hg[1]
            if d_pre is not None:
                grads[13], grads[14], grads[15], grads[16], grads[17], grads[18] = hg[2], hg[3], hg[4], hg[5], hg[6], hg[7]
            if config.FOLD_GAP_GRAD and ctx.sv1.gn is None and ctx.sv1.prelu is None and ops.bn_rowadd_ok(a1.shape[1], dt):
                row_g = d_g        # d a1 += d_g[n][c] / S: folded into the BatchNorm backward of ops.1, never materialised
            else:
                d_a1 = ops.gap_backward(d_g, a1, d_a1, dt)
        # deep-supervision head (pcrlv2_model_3d.py:60,71)
        if d_mask is not None:
            dx_ds, g_dw, g_db, g_dg, g_dbe = ops.luconv_backward(ctx.svd, d_mask, dsw, dsg, ld._packed, dt, need_dx=True, dx_add=d_a1)
            d_a1 = dx_ds
            grads[19], grads[20], grads[21], grads[22] = g_dw, g_db, g_dg, g_dbe
        # ---- ops.1, ops.0 ----
        # ops.0's activation has ops.1 as its only consumer: ops.1's data gradient takes the first pass of ops.0's BatchNorm backward (`bnred`)
        d_a0, gw1, gb1, gg1, gbe1 = ops.luconv_backward(ctx.sv1, d_a1, w1, g1, l1._packed, dt, need_dx=True, da_row_g=row_g, bnred=ctx.sv0)
        _park_slope(l1, ctx.sv1)
        grads[7], grads[8], grads[9], grads[10] = gw1, gb1, gg1, gbe1
        deferred = ()
        if ctx.sv0.kind == "upc":     # composed up_conv + conv1: both layers' gradients from one set of coarse-grid passes
            defer = config.DIRECT_PARAM_GRADS
            dx, g_upw, g_upb, gw0, gb0, gg0, gbe0 = ops.upconv_luconv_backward(ctx.sv0, d_a0, up_w, ctx.plist[1], w0, ctx.plist[3], g0,
                                                                               mod._composed_up, dt, need_dx=ctx.needs_input_grad[0],
                                                                               defer=defer)
            if defer:                 # delivered (and marked final) by _end_of_backward ...
                deferred = (0, 1, 2)
                _queue_end_of_backward()
                if getattr(ctx, "pass_idx", 1) == 0 and (config.EARLY_COMPOSED or _final_callback is not None):
                    # ... or right here when this is the step's first forward pass (its backward runs last -- the assumption mark_final
                    # already makes): the chain rule runs on the side stream next to the rest of the backward instead of as a serial tail at
                    # its end, and under the data-parallel wrapper the bucket's all-reduce overlaps the rest of the backward
                    _deliver_composed(mod._composed_up)
        else:
            g_upb = torch.empty(up_w.shape[1], dtype=torch.float32, device=d_a0.device)
            d_up, gw0, gb0, gg0, gbe0 = ops.luconv_backward(ctx.sv0, d_a0, w0, g0, l0._packed, dt, need_dx=True, dx_colsum=g_upb)
            _park_slope(l0, ctx.sv0)
            # ---- up_conv ----
            dx, g_upw, g_upb = ops.convt_backward(ctx.x, d_up, up_w, mod._packed_up, dt, need_dx=ctx.needs_input_grad[0], db=g_upb)
        grads[3], grads[4], grads[5], grads[6] = gw0, gb0, gg0, gbe0
        grads[0], grads[1], grads[2] = dx, g_upw, g_upb
        ctx.svd.x = None
        for k, p in enumerate(ctx.plist):
            grads[k + 1] = _park(p, grads[k + 1])
        mark_final(ctx, [p for k, p in enumerate(ctx.plist) if k not in deferred])
        return tuple(grads)


class OutFn(Function):
    """OutputTransition.forward: sigmoid(final_conv(x))  --  models/pcrlv2_model_3d.py:81-83."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        dt = mod.compute_dtype
        x = ops.to_act(x, dt)
        with ops.side_branch(x.device, x):      # nothing in the forward consumes the reconstruction: next to the next pass's encoder
            out = ops.conv1x1_to1_forward(x, w, b, dt)
        ctx.x, ctx.w, ctx.dt = x, w, dt
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.plist = (w, b)
        ctx.save_for_backward(out)   # an output: never stash it on ctx directly (reference cycle)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return None, None, None, None
        dx, dw, db = ops.conv1x1_to1_backward(ctx.x, ctx.saved_tensors[0], dout, ctx.w, ctx.dt, need_dx=ctx.needs_input_grad[0])
        out = dx, _park(ctx.plist[0], dw), _park(ctx.plist[1], db), None
        mark_final(ctx, ctx.plist)
        return out


class TrilinearFn(Function):
    """F.interpolate(x, scale_factor=s, mode='trilinear')  --  models/pcrlv2_model_3d.py:125-126."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.shape, ctx.scale = tuple(x.shape), scale
        ctx.set_materialize_grads(False)
        with ops.side_branch(x.device, x):      # behind the deep-supervision branch that produced x (UpStageFn.forward)
            return ops.upsample_forward(x, scale)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        return ops.upsample_backward(dy, ctx.shape, ctx.scale), None


class MSELossFn(Function):
    """nn.MSELoss()(p, gt)  --  train_3d.py:56,135,137."""

    @staticmethod
    def forward(ctx, p, gt):
        p, gt = p.contiguous().float(), gt.contiguous().float()
        ctx.p, ctx.gt = p, gt
        return ops.mse_forward(p, gt)

    @staticmethod
    def backward(ctx, dloss):
        return ops.mse_backward(ctx.p, ctx.gt, dloss), None


class CosineMeanFn(Function):
    """nn.CosineSimilarity(dim=1, eps=1e-8)(x, y.detach()).mean()  --  train_3d.py:57,90-91.
    Gradient flows to x only (the reference always detaches the second operand)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous().float(), y.contiguous().float()
        out, saved = ops.cosine_mean_forward(x, y)
        ctx.x, ctx.y, ctx.saved = x, y, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        return ops.cosine_mean_backward(ctx.x, ctx.y, ctx.saved, dout), None


class CosineTermsFn(Function):
    """Every cosine term of one step in one launch (train_3d.py:119-134): out[g] = sum_t w_t * mean_r cos(x_t[r], y_t[r].detach()).

    forward(ctx, spec, rows, ngroups, *tensors): `tensors` are float32 [R, C] matrices; spec = [(xi, x_row0, yi, y_row0, w, group), ...]
    names a term by the indices of its operands in `tensors` and the first of its `rows` rows (local views are row blocks of [6n, C]).
    Gradients flow to the x operands only; a tensor that is no term's x gets None (like an unused scale in the reference)."""

    @staticmethod
    def forward(ctx, spec, rows, ngroups, *tensors):
        import ctypes
        ts = [t.contiguous().float() for t in tensors]
        n = len(spec)
        P, F32, I32 = ctypes.c_void_p * n, ctypes.c_float * n, ctypes.c_int * n
        ptr = lambda i, r0: ts[i].data_ptr() + 4 * r0 * ts[i].shape[1]
        x, y = P(*[ptr(s[0], s[1]) for s in spec]), P(*[ptr(s[2], s[3]) for s in spec])
        w, C, grp = F32(*[float(s[4]) for s in spec]), I32(*[ts[s[0]].shape[1] for s in spec]), I32(*[int(s[5]) for s in spec])
        out = torch.empty(ngroups, dtype=torch.float32, device=ts[0].device)
        adr = ctypes.addressof
        L = ops.lib()
        nb = L.call("pcrl_cosine_terms_ws_bytes", n)
        L.call("pcrl_cosine_terms_fwd", adr(x), adr(y), adr(w), adr(C), adr(grp), n, rows, ngroups, 1e-8, out, ops.workspace(nb, out.device), nb,
               ops.stream_handle())
        ctx.ts, ctx.spec, ctx.rows, ctx.ngroups = ts, spec, rows, ngroups
        ctx.host = (x, y, w, C, grp)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        ts, spec, n = ctx.ts, ctx.spec, len(ctx.spec)
        x, y, w, C, grp = ctx.host
        grads, seen = [None] * len(ts), set()
        first = []
        for s in spec:
            key = (s[0], s[1])                      # a row block of a tensor: the first term that touches it stores, later ones add
            first.append(0 if key in seen else 1)
            seen.add(key)
        # row blocks of a multi-block tensor (the local views) that no term writes must read as zero: those tensors are carved out of ONE
        # zero-filled buffer (one fill launch for all of them instead of one per tensor)
        used = sorted({s[0] for s in spec})
        holes = [i for i in used if ts[i].shape[0] != ctx.rows and len({r0 for (xi, r0) in seen if xi == i}) * ctx.rows != ts[i].shape[0]]
        if holes:
            flat = torch.empty(sum(ts[i].numel() for i in holes), dtype=torch.float32, device=ts[0].device)
            ops.lib().call("pcrl_zero", flat, flat.numel() * 4, ops.stream_handle())
            o = 0
            for i in holes:
                grads[i] = flat[o:o + ts[i].numel()].view_as(ts[i])
                o += ts[i].numel()
        for i in used:
            if grads[i] is None:
                grads[i] = torch.empty_like(ts[i])
        P, I32 = ctypes.c_void_p * n, ctypes.c_int * n
        dx = P(*[grads[s[0]].data_ptr() + 4 * s[1] * ts[s[0]].shape[1] for s in spec])
        fst = I32(*first)
        adr = ctypes.addressof
        ops.lib().call("pcrl_cosine_terms_bwd", adr(x), adr(y), adr(dx), adr(w), adr(C), adr(grp), adr(fst), n, ctx.rows, ctx.ngroups, 1e-8,
                       dout.contiguous().float(), ops.stream_handle())
        return (None, None, None) + tuple(grads)


class LossTotalFn(Function):
    """loss = loss1 + loss2 + beta * l4 + local_loss (train_3d.py:136-138) from four device scalars in one launch -> (total, beta * l4).
    Backward: the four incoming scalars get g, g, beta * g, g (g = d total + nothing through the second output, which is only reported)."""

    @staticmethod
    def forward(ctx, l1, l2, l4, l5, beta):
        out = torch.empty(2, dtype=torch.float32, device=l1.device)
        f = lambda t: t.detach().reshape(1).float()
        ops.lib().call("pcrl_loss_total", f(l1), f(l2), f(l4), f(l5), float(beta), out, ops.stream_handle())
        ctx.beta = float(beta)
        total, scaled = out[0], out[1]
        ctx.mark_non_differentiable(scaled)
        return total, scaled

    @staticmethod
    def backward(ctx, g, _g_scaled):
        return g, g, g * ctx.beta, g, None


def loss_total(l1, l2, l4, l5, beta):
    return LossTotalFn.apply(l1, l2, l4, l5, beta)


class LossTailFn(Function):
    """LossTotalFn taking the two cosine groups as the [2] vector pcrl_cosine_terms_fwd wrote: (loss1, cos = [global, local], l4, beta) ->
    (total, beta * l4, global, local).  No select / select_backward / add_ / mul nodes between the cosine launch and the total: the backward is
    one launch (pcrl_loss_total_bwd) whose output is handed out as views -- g to loss1, beta * g to l4, [g, g] to the cosine terms."""

    @staticmethod
    def forward(ctx, l1, cos, l4, beta):
        ctx.set_materialize_grads(False)
        out = torch.empty(2, dtype=torch.float32, device=l1.device)
        f = lambda t: t.detach().reshape(1).float()
        c = cos.detach()
        ops.lib().call("pcrl_loss_total", f(l1), c[0:1], f(l4), c[1:2], float(beta), out, ops.stream_handle())
        ctx.beta = float(beta)
        total, scaled, lg, ll = out[0], out[1], c[0], c[1]
        ctx.mark_non_differentiable(scaled, lg, ll)
        return total, scaled, lg, ll

    @staticmethod
    def backward(ctx, g, _gs, _gg, _gl):
        if g is None:
            return None, None, None, None
        out = torch.empty(4, dtype=torch.float32, device=g.device)
        ops.lib().call("pcrl_loss_total_bwd", g.reshape(1).float(), ctx.beta, out, ops.stream_handle())
        return out[0], out[2:4], out[1], None


def loss_tail(l1, cos, l4, beta):
    return LossTailFn.apply(l1, cos, l4, beta)


_root_grads: dict = {}


def root_gradient(loss):
    """d loss / d loss = 1 as a cached device scalar: `loss.backward(gradient=...)` with it skips autograd's ones_like fill launch."""
    key = (str(loss.device), loss.dtype)
    t = _root_grads.get(key)
    if t is None:
        t = _root_grads[key] = torch.ones((), dtype=loss.dtype).to(loss.device)     # (built on the host: a copy, not a fill kernel)
    return t


def cosine_terms(spec, rows, ngroups, tensors):
    return CosineTermsFn.apply(spec, rows, ngroups, *tensors)


def mse_loss(p, gt):
    return MSELossFn.apply(p, gt)


def cosine_mean(x, y_detached):
    return CosineMeanFn.apply(x, y_detached.detach())


class NTXentFn(Function):
    """OPTIONAL EXTRA -- not the reference's loss (train_3d.py:86-92 is a negative cosine similarity with stop-gradient).
    NT-Xent / SimCLR: cross-entropy over cosine similarities / temperature of 2N embeddings, positives = the two views."""

    @staticmethod
    def forward(ctx, z, tau):
        z = z.contiguous().float()
        loss, ws = ops.ntxent_forward(z, tau)
        ctx.save_for_backward(z, ws)
        ctx.tau = tau
        return loss

    @staticmethod
    def backward(ctx, dloss):
        z, ws = ctx.saved_tensors
        return ops.ntxent_backward(z, dloss, ws, ctx.tau), None


def ntxent_loss(z1, z2, temperature=0.5):
    """NT-Xent over the 2N rows [z1; z2] (gradients flow to both).  Not used by train_3d (the reference has no such loss)."""
    return NTXentFn.apply(torch.cat([z1, z2], dim=0), float(temperature))


class GroupNormActFn(Function):
    """OPTIONAL EXTRA -- GroupNorm(groups) + activation (SiLU by default) on an NDHWC activation.  Not used by PCRLv23d: the
    reference instantiates BatchNorm3d + ReLU only and its own norm='gn' option crashes at construction."""

    @staticmethod
    def forward(ctx, y, gamma, beta, groups, act):
        a, saved = ops.gn_act_forward(y, gamma.detach(), beta.detach(), groups, act, y.dtype)
        ctx.save_for_backward(*saved, gamma)
        ctx.groups, ctx.act = groups, act
        return a

    @staticmethod
    def backward(ctx, da):
        *saved, gamma = ctx.saved_tensors
        dy, dg, db = ops.gn_act_backward(ops.to_act(da, saved[0].dtype), tuple(saved), gamma.detach(), ctx.groups, ctx.act, saved[0].dtype)
        return dy, dg, db, None, None


def group_norm_silu(y, gamma, beta, groups=8):
    """SiLU(GroupNorm(groups)(y)) for an activation in the engine's layout (ops.to_act / channels_last_3d)."""
    return GroupNormActFn.apply(y, gamma, beta, groups, ACT_SILU)

