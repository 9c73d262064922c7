"""autograd.Function wrappers of the 2D path: one Function per conv+BatchNorm2d+activation unit, the BasicBlock
tail, the pools / upsamplers and the projection + predictor heads of models/pcrlv2_model.py:108-127.

Parameter gradients are parked like the 3D stages do (functions._park): autograd only carries activation gradients.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import config, ops, ops2d
from ._lib import ACT_NONE, ACT_RELU, dtype_code, lib, stream_handle
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


HEAD_STATS_ONLY = True      # module attribute (the bit-identity test flips it): see DecoderBlockFn.forward
FUSED_ENCODER = __import__("os").environ.get("PCRL_FUSED_ENCODER_2D", "1") != "0"   # A/B switch (bit-identical results): 0 = one autograd node per unit
STEM_KERNEL = __import__("os").environ.get("PCRL_STEM_KERNEL_2D", "1") != "0"       # A/B switch: 0 = the general gather kernel on the image padded to 8 channels


class EncoderFn(Function):
    """smp ResNetEncoder('resnet18').forward (torchvision ResNet-18 without fc; models/pcrlv2_model.py:200) down to its LAST feature map --
    the only one the decoder reads (the skips are ignored, :115-117) -- as ONE autograd node.  As separate nodes every BasicBlock left
    autograd a two-way sum at its input (conv1 branch + identity branch): 24 full-tensor aten::add launches per C5 step, next to 24 add+relu
    and 24 mask passes.  Here: BatchNorm apply + identity add + ReLU in one pass (pcrl_bn_add_relu_fwd), the two gradients of a block's output
    summed inside the mask pass (pcrl_relu_mask_sum_bwd) or inside the max-pool backward (pcrl_maxpool2d_3s2_bwd_sum), and the stem's
    BatchNorm apply + ReLU + MaxPool2d(3, 2, 1) from one pass over the convolution output (the full-resolution stem activation is never
    stored).  Values are those of the per-unit chain bit for bit (config switch PCRL_FUSED_ENCODER_2D; tests/test_model2d_gpu.py).

    inputs : image (float32 NCHW), encoder module, then [conv.weight, bn.weight, bn.bias] of every unit in ResNetEncoder._units() order
    output : layer4's output (activation)"""

    @staticmethod
    def forward(ctx, x, enc, *params):
        units = enc._units()
        dt = units[0].compute_dtype
        L = lib()
        P = [params[3 * i:3 * i + 3] for i in range(len(units))]
        pi = {id(u): k for k, u in enumerate(units)}

        def conv_bn(xin, u):
            w, g, be = P[pi[id(u)]]
            y, partial, rows = ops2d.conv2d_forward(xin, w, None, u._packed, u.stride, u.pad, 0, dt)
            N, H, W, C = ops2d.dims2(y)
            bn = u.bn_module
            coef = ops.bn_finalize(partial, rows, C, N * H * W, g.detach(), be.detach(), bn.running_mean, bn.running_var)
            u._count_batch()
            return y, coef

        stem = enc._stem
        w0_, g0_, be0_ = P[0]
        if STEM_KERNEL and ops2d.stem_ok(x, w0_, dt):
            # the dedicated stem kernels read the loader's float32 NCHW image as it is (csrc/stem2d.hip): no padded bf16 copy of the image
            if getattr(stem, "_packed_stem", None) is None:
                stem._packed_stem = ops2d.PackedStem()
            img = x
            ctx.stem_kernel = True
            y0, partial, rows = ops2d.stem_forward(x, w0_, stem._packed_stem, dt)
            N, H, W, C = ops2d.dims2(y0)
            bn0 = stem.bn_module
            c0 = ops.bn_finalize(partial, rows, C, N * H * W, g0_.detach(), be0_.detach(), bn0.running_mean, bn0.running_var)
            stem._count_batch()
        else:
            ctx.stem_kernel = False
            img = ops2d.image_to_act(x, dt, 8)
            y0, c0 = conv_bn(img, stem)
        N, H, W, C = ops2d.dims2(y0)
        Ho, Wo = ops2d.out_size(H, 3, 2, 1), ops2d.out_size(W, 3, 2, 1)
        h = ops2d.new_act2(N, Ho, Wo, C, dt, y0.device)
        idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=y0.device)
        L.call("pcrl_bn_relu_maxpool2d_3s2_fwd", y0, c0[2], c0[3], h, idx, N, H, W, C, dtype_code(dt), stream_handle())
        saved = []
        for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
            for blk in layer:
                y1, c1 = conv_bn(h, blk._u1)
                N, H, W, C = ops2d.dims2(y1)
                M = N * H * W
                a1 = ops.bn_act_apply(y1, c1[2], c1[3], M, C, ACT_RELU, dt)
                y2, c2 = conv_bn(a1, blk._u2)
                out = torch.empty_like(y2)
                if blk._ud is not None:
                    yd, cd = conv_bn(h, blk._ud)
                    L.call("pcrl_bn_add_relu_fwd", y2, c2[2], c2[3], yd, cd[2], cd[3], out, M, C, dtype_code(dt), stream_handle())
                else:
                    yd = cd = None
                    L.call("pcrl_bn_add_relu_fwd", y2, c2[2], c2[3], h, None, None, out, M, C, dtype_code(dt), stream_handle())
                saved.append((blk, h, y1, c1, a1, y2, c2, yd, cd, M, C))
                h = out
        # the block outputs are needed by the mask passes: every one but the last is an intermediate (stashed as is); the last is this node's
        # OUTPUT and goes through save_for_backward (an output on ctx is a reference cycle, see functions.UpStageFn)
        ctx.outs = [s[1] for s in saved[1:]]
        ctx.stem = (img, y0, c0, idx, ops2d.dims2(y0))
        ctx.saved, ctx.units, ctx.P, ctx.pi, ctx.dt = saved, units, P, pi, dt
        ctx.pass_idx = getattr(units[0], "_pass_idx", 1)
        ctx.plist = tuple(params)
        ctx.save_for_backward(h)
        ctx.set_materialize_grads(False)
        return h

    @staticmethod
    def backward(ctx, d_h):
        n_in = 2 + len(ctx.plist)
        if d_h is None:
            return (None,) * n_in
        dt, L, P, pi = ctx.dt, lib(), ctx.P, ctx.pi
        (h_last,) = ctx.saved_tensors
        outs = ctx.outs + [h_last]            # outs[k] = output of block k
        grads = {}

        def put(u, dw, dg, db):
            k = pi[id(u)]
            grads[3 * k], grads[3 * k + 1], grads[3 * k + 2] = dw, dg, db

        def bn_conv_bwd(da, u, xin, y, c, M, C, act, need_dx=True, below=None, pre=None):
            # below: {"y", "c"} of the BatchNorm + ReLU layer whose activation xin is and which has no other consumer -- this layer's data gradient
            # may take that layer's first backward pass with it (ops2d.conv2d_backward); pre: such a pass already taken for `da`
            w, g, _ = P[pi[id(u)]]
            dy, dg, db = ops.bn_act_backward(da, y, g.detach(), c[0], c[1], c[2], c[3], M, C, act, dt, pre_partial=pre)
            dx, dw = ops2d.conv2d_backward(xin, dy, w, u._packed, u.stride, u.pad, 0, dt, need_dx=need_dx, below=below)
            put(u, dw, dg, db)
            return dx

        g = ops2d.relu_mask_backward(ops2d.to_act2(d_h, dt), h_last, dt)
        first, second = None, None
        for k in range(len(ctx.saved) - 1, -1, -1):
            blk, h_in, y1, c1, a1, y2, c2, yd, cd, M, C = ctx.saved[k]
            below = {"y": y1, "c": c1}       # a1 = relu(bn1(conv1(h_in))) feeds conv2 only (torchvision BasicBlock)
            d_a1 = bn_conv_bwd(g, blk._u2, a1, y2, c2, M, C, ACT_NONE, below=below)
            first = bn_conv_bwd(d_a1, blk._u1, h_in, y1, c1, M, C, ACT_RELU, pre=below.get("pre"))
            second = bn_conv_bwd(g, blk._ud, h_in, yd, cd, M, C, ACT_NONE) if blk._ud is not None else g
            if k > 0:
                gp = torch.empty_like(first)
                L.call("pcrl_relu_mask_sum_bwd", first, second, outs[k - 1], gp, first.numel(), dtype_code(dt), stream_handle())
                g = gp
        img, y0, c0, idx, (N, H, W, C) = ctx.stem
        da0 = ops2d.new_act2(N, H, W, C, dt, y0.device)
        L.call("pcrl_maxpool2d_3s2_bwd_sum", first, second, idx, da0, N, H, W, C, dtype_code(dt), stream_handle())
        if ctx.stem_kernel:      # the dedicated stem kernels ran forward (recorded there, not inferred from the saved tensor): their weight gradient
            u0 = ctx.units[0]
            w0_, g0_, _ = P[0]
            dy0, dg0, db0 = ops.bn_act_backward(da0, y0, g0_.detach(), c0[0], c0[1], c0[2], c0[3], N * H * W, C, ACT_RELU, dt)
            put(u0, ops2d.stem_wgrad(img, dy0, w0_, dt), dg0, db0)
        else:
            bn_conv_bwd(da0, ctx.units[0], img, y0, c0, N * H * W, C, ACT_RELU, need_dx=False)
        out = (None, None) + tuple(_park(p, grads.get(i)) for i, p in enumerate(ctx.plist))
        mark_final(ctx, ctx.plist)
        ctx.saved = ctx.outs = ctx.stem = None      # intermediates held on ctx (not saved tensors): released with the backward, not with the graph
        return out


def _heads_backward(d_pro, d_pre, heads, x_pro, params):
    """Backward of x_pro = bn(gap); x_pre = predictor_head(x_pro) (pcrlv2_model.py:124-127) down to the pooled vector (ops.heads_backward)."""
    return ops.heads_backward(d_pro, d_pre, heads, x_pro, params)


class DecoderBlockFn(Function):
    """DecoderBlock.forward (pcrlv2_model.py:113-128) as ONE autograd node: nearest x2 -> conv1 -> conv2 -> {pooled projection + predictor
    heads, deep-supervision head}.  The block's output has three consumers (the next block, its own deep-supervision head, the pooled
    heads); as separate nodes autograd summed their gradients with full-resolution element-wise adds (2.9 ms of ATen launches per C5
    step) -- here the BatchNorm backward of conv2 takes the parts as they are (pcrl_bn_act_bwd_*_sum: two tensors + the pooled branch's
    per-(sample, channel) term).
    `want_mask` False: the deep-supervision head runs its convolution and the statistics update of its BatchNorm2d only (the module's
    state after a step is the reference's); its activation, the 1x1 convolution to 3 channels and the map are not computed -- nothing
    reads them for the second view, the local views and the scales the first cos_loss did not draw (train_2d.py:143-168).

    inputs : x, [w, gamma, beta] of conv1, conv2; [w, b, gamma, beta] of deep_supervision_head[0..1], [w, b] of [3]; bn.(gamma, beta),
             ph0.(w, b), ph1.(gamma, beta), ph3.(w, b); module, want_mask
    outputs: a2 (activation), x_pro [N,C], x_pre [N,C], x_mask float32 [N,3,H,W] (NHWC memory) | None"""

    @staticmethod
    def forward(ctx, x, w1, g1, be1, w2, g2, be2, wd, bd, gd, bed, w3, b3, bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b, mod, want_mask):
        dt = mod.compute_dtype
        u1, u2, ud0, ud3 = mod._u1, mod._u2, mod._ud0, mod._ud3
        dev, L = x.device, lib()

        def conv_bn(xin, w, b, g, be, u, stats_only=False):
            y, partial, rows = ops2d.conv2d_forward(xin, w, b, u._packed, 1, 1, u.up, dt, stats_only=stats_only)
            N, H, W = xin.shape[0], xin.shape[2] * (2 if u.up else 1), xin.shape[3] * (2 if u.up else 1)      # 3x3 / stride 1 / pad 1: the output grid is the (upsampled) input's
            C = w.shape[0]
            bn = u.bn_module
            coef = ops.bn_finalize(partial, rows, C, N * H * W, g.detach(), be.detach(), bn.running_mean, bn.running_var)
            u._count_batch()
            return y, coef

        y1, c1 = conv_bn(x, w1, None, g1, be1, u1)
        N, H, W, C = ops2d.dims2(y1)
        M = N * H * W
        a1 = ops.bn_act_apply(y1, c1[2], c1[3], M, C, ACT_RELU, dt)
        y2, c2 = conv_bn(a1, w2, None, g2, be2, u2)
        if config.FUSE_APPLY_CONSUMERS and ops.bn_rowadd_ok(C, dt):      # the activation and its global average pool from one pass
            a2, g = torch.empty_like(y2), ops._f32(N * C, dev).view(N, C)
            nbg = L.call("pcrl_gap_ws_bytes", N, H * W, C)
            L.call("pcrl_bn_act_apply_gap", y2, a2, g, c2[2], c2[3], ops.workspace(nbg, dev), nbg, N, H * W, C, ACT_RELU, dtype_code(dt), stream_handle())
        else:
            a2 = ops.bn_act_apply(y2, c2[2], c2[3], M, C, ACT_RELU, dt)
            g = ops2d.gap_forward(a2, dt)
        x_pro, m_pro, r_pro = ops.bn1d_forward(g, bn_g, bn_b, mod.bn.running_mean, mod.bn.running_var, relu=False)
        h0 = ops.linear_forward(x_pro, p0_w, p0_b)
        ph1 = mod.predictor_head[1]
        h1, m_h, r_h = ops.bn1d_forward(h0, p1_g, p1_b, ph1.running_mean, ph1.running_var, relu=True)
        x_pre = ops.linear_forward(h1, p3_w, p3_b)
        mod._count_batch_heads()
        # statistics (running_mean / running_var / num_batches_tracked) always; without a reader (want_mask False) the convolution's output is not
        # even written where the kernel can leave it out (HEAD_STATS_ONLY: the 16- and 32-channel heads at 512^2 / 256^2, 0.8 GB per global view)
        yd, cd = conv_bn(a2, wd, bd, gd, bed, ud0, stats_only=HEAD_STATS_ONLY and not want_mask)
        ad = x_mask = None
        if want_mask:
            ad = ops.bn_act_apply(yd, cd[2], cd[3], M, C, ACT_RELU, dt)
            x_mask = ops2d.conv2d_forward(ad, w3, b3, ud3._packed, 1, 0, 0, dt, want_stats=False, out_f32=True)[0]
        ctx.mod, ctx.dt, ctx.MC = mod, dt, (N, H * W, M, C)
        ctx.x, ctx.l1, ctx.l2 = x, (y1, c1, a1), (y2, c2)
        ctx.ld = (yd, cd, ad) if want_mask else None
        ctx.heads = (g, m_pro, r_pro, h0, h1, m_h, r_h)
        ctx.plist = (w1, g1, be1, w2, g2, be2, wd, bd, gd, bed, w3, b3, bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b)
        ctx.pass_idx = getattr(mod, "_pass_idx", 1)
        ctx.save_for_backward(a2, x_pro)      # OUTPUTS needed in backward: never stashed on ctx directly (reference cycle, see functions.UpStageFn)
        ctx.set_materialize_grads(False)
        return a2, x_pro, x_pre, x_mask

    @staticmethod
    def backward(ctx, d_a2, d_pro, d_pre, d_mask):
        n_in = 23
        if d_a2 is None and d_pro is None and d_pre is None and d_mask is None:
            return (None,) * n_in
        mod, dt = ctx.mod, ctx.dt
        N, S, M, C = ctx.MC
        w1, g1, be1, w2, g2, be2, wd, bd, gd, bed, w3, b3, bn_g, bn_b, p0_w, p0_b, p1_g, p1_b, p3_w, p3_b = ctx.plist
        a2, x_pro = ctx.saved_tensors
        u1, u2, ud0 = mod._u1, mod._u2, mod._ud0
        pg = [None] * 20         # parameter gradients in ctx.plist order
        row_g = None
        if d_pro is not None or d_pre is not None:
            row_g, hg = _heads_backward(d_pro, d_pre, ctx.heads, x_pro, ctx.plist[12:])
            pg[12:20] = hg
        dx_ds = None
        if d_mask is not None:
            if ctx.ld is None:
                raise RuntimeError("DecoderBlockFn: a gradient arrived for a deep-supervision map that was not computed")
            yd, cd, ad = ctx.ld
            d_ad, g_w3, g_b3 = ops2d.conv1x1_small_backward(ad, ops2d.to_act2(d_mask, torch.float32), w3, dt)
            dyd, g_gd, g_bed = ops.bn_act_backward(d_ad, yd, gd.detach(), cd[0], cd[1], cd[2], cd[3], M, C, ACT_RELU, dt)
            dx_ds, g_wd = ops2d.conv2d_backward(a2, dyd, wd, ud0._packed, 1, 1, 0, dt, need_dx=True)
            pg[6], pg[7], pg[8], pg[9], pg[10], pg[11] = g_wd, ops.zero_grad_vector(C, a2.device), g_gd, g_bed, g_w3, g_b3
        da = ops2d.to_act2(d_a2, dt) if d_a2 is not None else None
        da2 = dx_ds
        if da is None:
            da, da2 = da2, None
        dx = None
        if da is not None or row_g is not None:
            y2, c2 = ctx.l2
            y1, c1, a1 = ctx.l1
            if row_g is not None and not ops.bn_rowadd_ok(C, dt):
                da, row_g = ops2d.gap_backward(row_g, a2, dt, add_src=da), None
            if da2 is not None and not ops.bn_rowadd_ok(C, dt):
                da, da2 = da + da2, None
            dy2, g_g2, g_be2 = ops.bn_act_backward(da, y2, g2.detach(), c2[0], c2[1], c2[2], c2[3], M, C, ACT_RELU, dt, row_g=row_g, da2=da2)
            below = {"y": y1, "c": c1}       # a1 = relu(bn(conv1(up(x)))) feeds conv2 only (models/pcrlv2_model.py:115-118)
            d_a1, g_w2 = ops2d.conv2d_backward(a1, dy2, w2, u2._packed, 1, 1, 0, dt, need_dx=True, below=below)
            dy1, g_g1, g_be1 = ops.bn_act_backward(d_a1, y1, g1.detach(), c1[0], c1[1], c1[2], c1[3], M, C, ACT_RELU, dt, pre_partial=below.get("pre"))
            dx, g_w1 = ops2d.conv2d_backward(ctx.x, dy1, w1, u1._packed, 1, 1, 1, dt, need_dx=ctx.needs_input_grad[0])
            pg[0], pg[1], pg[2], pg[3], pg[4], pg[5] = g_w1, g_g1, g_be1, g_w2, g_g2, g_be2
        out = (dx,) + tuple(_park(p, g) for p, g in zip(ctx.plist, pg)) + (None, None)
        mark_final(ctx, ctx.plist)
        ctx.x = ctx.l1 = ctx.l2 = ctx.ld = ctx.heads = None      # per-block activations held on ctx: released with the backward
        return out


class SegMSEFn(Function):
    """criterion(self.model.segmentation_head(decoder_output), gt) -- pcrlv2_model.py:208 + train_2d.py:165 -- as one node: Conv2d(16, n_class,
    3, padding=1) with float32 NHWC output, the MSE against the NCHW image read in place; the backward writes the 3-channel gradient
    directly in the zero-padded form the convolution backward kernels take and its bias sums in the same pass."""

    @staticmethod
    def forward(ctx, h, w, b, gt, unit):
        dt = unit.compute_dtype
        y = ops2d.conv2d_forward(h, w, b, unit._packed, unit.stride, unit.pad, 0, dt, want_stats=False, out_f32=True)[0]
        gt = ops2d._nchw_f32(gt, h.device)
        loss = ops2d.mse2d_forward(y, gt)
        ctx.h, ctx.y, ctx.gt, ctx.unit, ctx.dt = h, y, gt, unit, dt
        ctx.plist = (w, b)
        ctx.pass_idx = getattr(unit, "_pass_idx", 1)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        unit, dt = ctx.unit, ctx.dt
        w, b = ctx.plist
        Co = w.shape[0]
        CoP = ops2d._pow2_at_least_8(Co)
        dy, colpart, rows = ops2d.mse2d_backward(ctx.y, ctx.gt, dloss, CoP, dt)
        db = ops2d.colsum_f32(colpart, rows, CoP)[:Co] if b is not None else None
        dx, dw = ops2d.conv2d_backward(ctx.h, dy, w, unit._packed, unit.stride, unit.pad, 0, dt, need_dx=ctx.needs_input_grad[0])
        out = dx, _park(w, dw), (_park(b, db) if b is not None else None), None, None
        mark_final(ctx, [p for p in ctx.plist if p is not None])
        return out


class MaskMSEFn(Function):
    """criterion(F.interpolate(x_mask, scale_factor=s, mode='bilinear'), gt) -- pcrlv2_model.py:190 + train_2d.py:167 -- for the ONE
    deep-supervision map a step uses: bilinear upsampling (skipped at s = 1), the MSE against the NCHW image, and back."""

    @staticmethod
    def forward(ctx, x_mask, gt, scale):
        x = ops2d.to_act2(x_mask, torch.float32)
        up = ops2d.bilinear_forward(x, scale) if scale != 1 else x
        gt = ops2d._nchw_f32(gt, x.device)
        ctx.in_dims, ctx.scale, ctx.up, ctx.gt = ops2d.dims2(x), scale, up, gt
        return ops2d.mse2d_forward(up, gt)

    @staticmethod
    def backward(ctx, dloss):
        C = ctx.in_dims[3]
        d_up, _colpart, _rows = ops2d.mse2d_backward(ctx.up, ctx.gt, dloss, C, torch.float32)
        d = ops2d.bilinear_backward(d_up, ctx.in_dims, ctx.scale) if ctx.scale != 1 else d_up
        return d, None, None


def mse_loss2d(p, gt):
    """nn.MSELoss()(p, gt) for an NHWC-memory prediction [N,C,H,W] and a target in any layout (train_2d.py:165,167)."""
    from .functions import MSELossFn
    return MSELossFn.apply(p.permute(0, 2, 3, 1), gt.to(device=p.device, dtype=torch.float32).permute(0, 2, 3, 1))
