"""bfloat16-ROUNDING-AWARE comparator for the engine's `--amp` mode.  TEST INFRASTRUCTURE ONLY (tests/ and oracle/make_emulated.py).

`pcrlv2_oracle` is the reference's algorithm in exact (float64) arithmetic; the engine's bfloat16 mode differs from it by the roundings
it performs, and a bf16 step compared with the float64 golden can only be held to what those roundings do to an ill-conditioned quantity
(gradient norms to 25 %, directions to 0.75 in tests/test_model_gpu.py).  This module is the SAME algorithm in float64 with a
round-to-nearest-even bfloat16 rounding inserted exactly where the engine rounds:

  * convolution operands: packed weights (pcrl_pack_conv3_weight), the composed up-conv weights (rounded up_conv / conv1 weights enter the
    composition, the composed weights are rounded again; upconv_fused.hip), the network input of the first layer;
  * every stored activation: the convolution output `y` (rounded AFTER the batch statistics were taken from the float32 accumulators) and
    the normalised / activated tensor `a`;
  * in backward the same tensors' gradients (`da`, `dy`, `dx` are stored in bfloat16; weight gradients accumulate in float32 and are not
    rounded), and the gradient of the composed weights before the chain rule (upconv_fused.hip: pcrl_upconv_wgrad_finish);
  * NOT rounded (float32 in the engine): statistics, heads (global average, BatchNorm1d, Linear), 1-channel maps, losses, parameters.

What is left between the engine and this comparator is float32-vs-float64 accumulation and the places where the emulation is not exact
(the order in which three gradient contributions to a decoder stage's output are rounded) -- an order of magnitude less than the
roundings themselves, which is what lets the tests hold gradient norms to 3 % and directions to 0.97.

The composed operator follows DESIGN.md 4.5 / tests/test_host_cpu.py::test_composed_upconv_algebra_in_float64."""
from __future__ import annotations

import itertools
import math
import random

import torch
import torch.nn.functional as F

import pcrlv2_oracle as O


class _RoundBoth(torch.autograd.Function):
    """value -> bf16 -> value; the gradient passing back is rounded too (the engine stores this tensor AND its gradient in bfloat16)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundValue(torch.autograd.Function):
    """value -> bf16 -> value, identity derivative (an operand that is rounded on its way into the matrix pipe; its gradient is float32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


rb, rv = _RoundBoth.apply, _RoundValue.apply

# Memory: float64 autograd keeps ~6 full tensors per layer; with CHECKPOINT on every LUConv / decoder stage is recomputed in backward
# (torch.utils.checkpoint, non-reentrant) and only its input is kept -- what lets the b = 8, 64x64x32 case fit a 62 GB host.  The running
# statistics are not tracked then (new_bufs must be None: a recomputation would update them twice).
CHECKPOINT = False


def _ckpt(fn, *args):
    if not CHECKPOINT:
        return fn(*args)
    from torch.utils.checkpoint import checkpoint
    return checkpoint(fn, *args, use_reentrant=False)

_PAIRS = {(0, 0): [(0, 1)], (0, 1): [(1, 0), (2, 1)], (1, 0): [(0, 0), (1, 1)], (1, 1): [(2, 0)]}   # per axis: (p, q) -> [(t, s)]


def _bn_apply(y_full, y, st, p, new_bufs):
    """Training-mode BatchNorm whose statistics come from the UNROUNDED accumulators `y_full` and are applied to the stored tensor `y`."""
    dims = [0] + list(range(2, y.dim()))
    m = y.numel() // y.shape[1]
    mean, var = y_full.mean(dim=dims), y_full.var(dim=dims, unbiased=False)
    shp = [1, -1] + [1] * (y.dim() - 2)
    out = (y - mean.view(shp)) / torch.sqrt(var.view(shp) + O.BN_EPS) * st[p + ".weight"].view(shp) + st[p + ".bias"].view(shp)
    if new_bufs is not None:
        with torch.no_grad():
            rm = new_bufs.get(p + ".running_mean", st[p + ".running_mean"])
            rvar = new_bufs.get(p + ".running_var", st[p + ".running_var"])
            nb = new_bufs.get(p + ".num_batches_tracked", st[p + ".num_batches_tracked"])
            new_bufs[p + ".running_mean"] = (1 - O.BN_MOMENTUM) * rm + O.BN_MOMENTUM * mean.detach().to(rm.dtype)
            new_bufs[p + ".running_var"] = (1 - O.BN_MOMENTUM) * rvar + O.BN_MOMENTUM * (var.detach() * (m / max(m - 1, 1))).to(rvar.dtype)
            new_bufs[p + ".num_batches_tracked"] = nb + 1
    return out


def _luconv(x, st, p, new_bufs, first=False):
    """conv3x3x3 (bf16 operands, exact accumulation) -> statistics -> y stored in bf16 -> BatchNorm + ReLU -> a stored in bf16."""
    if first:
        x = rv(x)       # the first layer's scalar field enters the matrix pipe as bfloat16
    y_full = F.conv3d(x, rv(st[p + ".conv1.weight"]), st[p + ".conv1.bias"], padding=1)
    return rb(torch.relu(_bn_apply(y_full, rb(y_full), st, p + ".bn1", new_bufs)))


def _to1_head(a, st, p, new_bufs):
    """Deep-supervision head: C -> 1 convolution on the bf16 activation with bf16 weights, float32 map from there on (no rounding)."""
    y = F.conv3d(a, rv(st[p + ".conv1.weight"]), st[p + ".conv1.bias"], padding=1)
    return torch.sigmoid(_bn_apply(y, y, st, p + ".bn1", new_bufs))


def _composed(x, w_up, b_up, w0, b0):
    """y0 = conv1(up_conv(x)) as the engine computes it: 8 phases x 8 taps of the composed, bf16-rounded weights on the zero-padded coarse
    tensor, plus the float32 bias field (b0 + the taps of w0 that stay inside the fine grid applied to b_up)."""
    N, Ci, D, H, W = x.shape
    Co = w0.shape[0]
    wu, wc = rv(w_up), rv(w0)
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    y = x.new_zeros(N, Co, 2 * D, 2 * H, 2 * W)
    for p in itertools.product((0, 1), repeat=3):
        acc = 0
        for q in itertools.product((0, 1), repeat=3):
            weff = 0
            for (td, sd), (th, sh), (tw, sw) in itertools.product(_PAIRS[p[0], q[0]], _PAIRS[p[1], q[1]], _PAIRS[p[2], q[2]]):
                weff = weff + wu[:, :, sd, sh, sw] @ wc[:, :, td, th, tw].t()
            weff = rb(weff)     # composed weights stored in bf16; their gradient is rounded before the chain rule as well
            sl = xp[:, :, p[0] + q[0]:p[0] + q[0] + D, p[1] + q[1]:p[1] + q[1] + H, p[2] + q[2]:p[2] + q[2] + W]
            acc = acc + torch.einsum("ncdhw,co->nodhw", sl, weff)
        y[:, :, p[0]::2, p[1]::2, p[2]::2] = acc
    # bias field: the same for every sample; float32 parameters, unrounded (upc_bias_kernel)
    field = F.conv3d(b_up.view(1, -1, 1, 1, 1).expand(1, -1, 2 * D, 2 * H, 2 * W), w0, b0, padding=1)
    return y + field


def _up_transition(x, st, name, new_bufs):
    b = x.shape[0]
    y_full = _composed(x, st[name + ".up_conv.weight"], st[name + ".up_conv.bias"], st[name + ".ops.0.conv1.weight"], st[name + ".ops.0.conv1.bias"])
    a0 = rb(torch.relu(_bn_apply(y_full, rb(y_full), st, name + ".ops.0.bn1", new_bufs)))
    a1 = _luconv(a0, st, name + ".ops.1", new_bufs)
    x_pro = a1.mean(dim=(2, 3, 4)).view(b, -1)
    x_pro = O._bn_train(x_pro, st, name + ".bn", new_bufs)
    h = F.linear(x_pro, st[name + ".predictor_head.0.weight"], st[name + ".predictor_head.0.bias"])
    h = torch.relu(O._bn_train(h, st, name + ".predictor_head.1", new_bufs))
    x_pre = F.linear(h, st[name + ".predictor_head.3.weight"], st[name + ".predictor_head.3.bias"])
    x_mask = _to1_head(a1, st, name + ".deep_supervision_head", new_bufs)
    return a1, x_pro, x_pre, x_mask


def forward(st, x, local=False, new_bufs=None):
    """pcrlv2_oracle.forward with the engine's bfloat16 rounding points (training mode only)."""
    h = x
    for i, (p, _, _) in enumerate(O.ENCODER):
        if i in (2, 4, 6):
            h = F.max_pool3d(h, 2)
        assert not (CHECKPOINT and new_bufs is not None), "running statistics cannot be tracked under checkpointing"
        h = _ckpt(lambda t, p=p, i=i: _luconv(t, st, p, new_bufs, first=(i == 0)), h)
    feats, masks_raw = [], []
    for name, _, _ in O.DECODER:
        h, pro, pre, mk = _ckpt(lambda t, name=name: _up_transition(t, st, name, new_bufs), h)
        feats.append([pro, pre])
        masks_raw.append(mk)
    masks = []
    if not local:
        masks.append(F.interpolate(masks_raw[0], scale_factor=4, mode="trilinear"))
        masks.append(F.interpolate(masks_raw[1], scale_factor=2, mode="trilinear"))
        masks.append(masks_raw[2])
    out = torch.sigmoid(F.conv3d(h, st["out_tr.final_conv.weight"], st["out_tr.final_conv.bias"]))   # 1x1x1: float32 weights (to1_fwd_kernel)
    return out, feats, masks


def step_losses(st, batch, epoch, rng: random.Random, new_bufs=None):
    """pcrlv2_oracle.step_losses on the rounding-aware forward (train_3d.py:109-138)."""
    input1, input2, gt, _gt2, local_views = batch
    bsz = input1.size(0)
    mask1, dec1, mid1 = forward(st, input1, new_bufs=new_bufs)
    _mask2, dec2, _ = forward(st, input2, new_bufs=new_bufs)
    loss2, index2 = O.cos_loss(dec1, dec2, rng)
    _, lout, _ = forward(st, torch.cat(local_views, dim=0), local=True, new_bufs=new_bufs)
    lout = [torch.stack(t) for t in lout]
    local_loss = 0.0
    for i in range(len(local_views)):
        tmp = [t[:, bsz * i: bsz * (i + 1)] for t in lout]
        l1, _ = O.cos_loss(dec1, tmp, rng)
        l2, _ = O.cos_loss(dec2, tmp, rng)
        local_loss = local_loss + l1 + l2
    local_loss = local_loss / (2 * len(local_views))
    loss1 = F.mse_loss(mask1, gt)
    beta = 0.5 * (1.0 + math.cos(math.pi * epoch / 240))
    loss4 = beta * F.mse_loss(mid1[index2], gt)
    loss = loss1 + loss2 + loss4 + local_loss
    return dict(loss=loss, loss1=loss1, loss2=loss2, loss4=loss4, local_loss=local_loss, index2=index2, mask1=mask1, dec1=dec1, mid1=mid1)
