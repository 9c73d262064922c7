"""autograd.Function wrappers of the 2D path: one Function per conv+BatchNorm2d+activation unit, the BasicBlock
tail, the pools / upsamplers and the projection + predictor heads of models/pcrlv2_model.py:108-127.

Parameter gradients are parked like the 3D stages do (functions._park): autograd only carries activation gradients.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops, ops2d
from .functions import _park, mark_final


def _grad_act(g, dtype, pad_to=0):
    return ops2d.to_act2(g, dtype, pad_to)


class ConvBNActFn(Function):
    """act(BatchNorm2d(conv2d(x)))  --  smp `Conv2dReLU` (pcrlv2_model.py:78-92), torchvision BasicBlock conv/bn pairs, the ResNet
    stem, the downsample branch (act = none) and deep_supervision_head.{0,1,2} (pcrlv2_model.py:103-105; conv with bias)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, mod):
        dt = mod.compute_dtype
        y, partial, rows = ops2d.conv2d_forward(x, w, b, mod._packed, mod.stride, mod.pad, mod.up, dt)
        N, Ho, Wo, Co = ops2d.dims2(y)
        M = N * Ho * Wo
        bn = mod.bn_module
        mean, rstd, scale, shift = ops.bn_finalize(partial, rows, Co, M, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var)
        a = ops.bn_act_apply(y, scale, shift, M, Co, mod.act, dt)
        mod._count_batch()
        ctx.x, ctx.y, ctx.coef = x, y, (mean, rstd, scale, shift)
        ctx.mod, ctx.dt, ctx.M = mod, dt, M
        ctx.plist = (w, b, gamma, beta)
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.set_materialize_grads(False)
        return a

    @staticmethod
    def backward(ctx, da):
        if da is None:
            return (None,) * 6
        mod, dt = ctx.mod, ctx.dt
        w, b, gamma, beta = ctx.plist
        mean, rstd, scale, shift = ctx.coef
        Co = w.shape[0]
        da = _grad_act(da, dt)
        dy, dgamma, dbeta = ops.bn_act_backward(da, ctx.y, gamma.detach(), mean, rstd, scale, shift, ctx.M, Co, mod.act, dt)
        dx, dw = ops2d.conv2d_backward(ctx.x, dy, w, mod._packed, mod.stride, mod.pad, mod.up, dt, need_dx=ctx.needs_input_grad[0])
        db = None
        if b is not None:
            db = ops.zero_grad_vector(Co, dy.device)   # a bias in front of batch statistics has an identically zero gradient
        out = dx, _park(w, dw), (_park(b, db) if b is not None else None), _park(gamma, dgamma), _park(beta, dbeta), None
        mark_final(ctx, [p for p in ctx.plist if p is not None])
        return out


class ConvFn(Function):
    """conv2d(x) + bias with float32 output: deep_supervision_head.3 (conv1x1 -> 3, pcrlv2_model.py:106) and the smp
    SegmentationHead conv3x3(16 -> n_class)."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        dt = mod.compute_dtype
        y, _, _ = ops2d.conv2d_forward(x, w, b, mod._packed, mod.stride, mod.pad, 0, dt, want_stats=False, out_f32=True)
        ctx.x, ctx.mod, ctx.dt = x, mod, dt
        ctx.plist = (w, b)
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None
        mod, dt = ctx.mod, ctx.dt
        w, b = ctx.plist
        Co = w.shape[0]
        CoP = ops2d._pow2_at_least_8(Co)
        dyf = ops2d.to_act2(dy, torch.float32, pad_to=CoP)      # 3-channel gradient zero-padded to 8 channels
        N, Ho, Wo, _ = ops2d.dims2(dyf)
        db = ops2d.colsum(dyf, N * Ho * Wo, CoP, torch.float32)[:Co] if b is not None else None
        dyp = ops2d.to_act2(dyf, dt)
        dx, dw = ops2d.conv2d_backward(ctx.x, dyp, w, mod._packed, mod.stride, mod.pad, 0, dt, need_dx=ctx.needs_input_grad[0])
        out = dx, _park(w, dw), (_park(b, db) if b is not None else None), None
        mark_final(ctx, [p for p in ctx.plist if p is not None])
        return out


class AddReluFn(Function):
    """relu(t + identity) of torchvision's BasicBlock."""

    @staticmethod
    def forward(ctx, t, r, dt):
        a = ops2d.add_relu_forward(t, ops2d.to_act2(r, dt), dt)
        ctx.dt = dt
        ctx.save_for_backward(a)
        ctx.set_materialize_grads(False)
        return a

    @staticmethod
    def backward(ctx, da):
        if da is None:
            return None, None, None
        (a,) = ctx.saved_tensors
        g = ops2d.relu_mask_backward(_grad_act(da, ctx.dt), a, ctx.dt)
        return g, g, None


class MaxPool2dFn(Function):
    """nn.MaxPool2d(3, 2, 1) of the ResNet stem."""

    @staticmethod
    def forward(ctx, x, dt):
        y, idx = ops2d.maxpool_forward(x, dt)
        ctx.idx, ctx.in_dims, ctx.dt = idx, ops2d.dims2(x), dt
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        return ops2d.maxpool_backward(_grad_act(dy, ctx.dt), ctx.idx, ctx.in_dims, ctx.dt), None


class BilinearFn(Function):
    """F.interpolate(x_mask, scale_factor=2 ** (4 - i), mode='bilinear')  --  pcrlv2_model.py:190."""

    @staticmethod
    def forward(ctx, x, scale):
        x = ops2d.to_act2(x, torch.float32)
        ctx.in_dims, ctx.scale = ops2d.dims2(x), scale
        ctx.set_materialize_grads(False)
        return ops2d.bilinear_forward(x, scale)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        return ops2d.bilinear_backward(ops2d.to_act2(dy, torch.float32), ctx.in_dims, ctx.scale), None


class ProjHeadFn(Function):
    """x_pro = bn(adaptive_avg_pool2d(x).view(b, -1)); x_pre = predictor_head(x_pro)  --  pcrlv2_model.py:124-127."""

    @staticmethod
    def forward(ctx, a, bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b, mod):
        dt = mod.compute_dtype
        g = ops2d.gap_forward(a, dt)
        x_pro, m_pro, r_pro = ops.bn1d_forward(g, bn_g, bn_b, mod.bn.running_mean, mod.bn.running_var, relu=False)
        h0 = ops.linear_forward(x_pro, p0_w, p0_b)
        ph1 = mod.predictor_head[1]
        h1, m_h, r_h = ops.bn1d_forward(h0, p1_g, p1_b, ph1.running_mean, ph1.running_var, relu=True)
        x_pre = ops.linear_forward(h1, p3_w, p3_b)
        mod._count_batch_heads()
        ctx.a, ctx.dt = a, dt
        ctx.heads = (g, m_pro, r_pro, h0, h1, m_h, r_h)
        ctx.plist = (bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b)
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.save_for_backward(x_pro)
        ctx.set_materialize_grads(False)
        return x_pro, x_pre

    @staticmethod
    def backward(ctx, d_pro, d_pre):
        if d_pro is None and d_pre is None:
            return (None,) * 10
        bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b = ctx.plist
        (x_pro,) = ctx.saved_tensors
        g, m_pro, r_pro, h0, h1, m_h, r_h = ctx.heads
        grads = [None] * 8
        d_xpro = d_pro.contiguous() if d_pro is not None else None
        if d_pre is not None:
            d_h1, g_p3w, g_p3b = ops.linear_backward(d_pre, h1, p3_w)
            d_h0, g_p1g, g_p1b = ops.bn1d_backward(d_h1, h0, h1, p1_g, m_h, r_h, relu=True)
            d_xp, g_p0w, g_p0b = ops.linear_backward(d_h0, x_pro, p0_w)
            d_xpro = d_xp if d_xpro is None else d_xpro + d_xp
            grads[2:8] = [g_p0w, g_p0b, g_p1g, g_p1b, g_p3w, g_p3b]
        d_g, g_bng, g_bnb = ops.bn1d_backward(d_xpro, g, x_pro, bn_g, m_pro, r_pro, relu=False)
        grads[0], grads[1] = g_bng, g_bnb
        d_a = ops2d.gap_backward(d_g, ctx.a, ctx.dt)
        out = (d_a,) + tuple(_park(p, gr) for p, gr in zip(ctx.plist, grads)) + (None,)
        mark_final(ctx, ctx.plist)
        return out


def mse_loss2d(p, gt):
    """nn.MSELoss()(p, gt) for an NHWC-memory prediction [N,C,H,W] and a target in any layout (train_2d.py:165,167)."""
    from .functions import MSELossFn
    return MSELossFn.apply(p.permute(0, 2, 3, 1), gt.to(device=p.device, dtype=torch.float32).permute(0, 2, 3, 1))
