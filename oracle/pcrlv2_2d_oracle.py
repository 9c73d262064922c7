"""CPU oracle for the PCRLv2 2D (ResNet-18 U-Net) pre-training path.  TEST INFRASTRUCTURE ONLY  (SURVEY 8f N1).

Functional restatement in plain PyTorch-CPU ops, driven by a state_dict with the reference's key names.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product path never does.

Parity status: **PARITY UNPINNED**.  The reference's 2D model cannot be constructed here: it subclasses
`segmentation_models_pytorch` (PyPI, version un-pinned: only named in README.md:7) which builds on torchvision -- both absent
from this image, and the reference holds no golden vectors or known-answer tests for this path.  Therefore:
  * the decoder, the heads and the loss assembly follow the reference line by line
    (models/pcrlv2_model.py:68-128 DecoderBlock, :131-194 PCRLv2Decoder, :197-209 PCRLv2; train_2d.py:111-171);
  * `md.Conv2dReLU(use_batchnorm=True)` is restated from smp's published definition: Conv2d(bias=False) -> BatchNorm2d -> ReLU;
    `md.Attention(None)` = identity; `SegmentationHead(16, n_class, kernel_size=3)` = Conv2d(16, n_class, 3, padding=1), no
    upsampling, no activation;
  * the encoder restates torchvision's published ResNet-18 (BasicBlock x [2,2,2,2], stem conv7x7 s2 + BN + ReLU + maxpool 3x3 s2,
    downsample = conv1x1(stride) + BN) as wrapped by smp's ResNetEncoder (features = [x, stem, layer1..layer4]).
Layout: logical NCHW; dtype follows the inputs (float64 or float32).
"""
from __future__ import annotations

import math
import random

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
DECODER_CHANNELS = (256, 128, 64, 32, 16)          # pcrlv2_model.py:137
BETA_PERIOD = 240                                  # train_2d.py:166


EVAL = False      # True (set by model_forward(..., training=False)): nn.Module.eval() -- every BatchNorm uses its running statistics, nothing is updated


def _bn(x, sd, prefix, state_out=None):
    """nn.BatchNorm{1,2}d in training mode: batch statistics (biased variance) + running-stat update (unbiased); in eval mode (EVAL) the running
    statistics, no update."""
    if EVAL:
        shape = [1, -1] + [1] * (x.dim() - 2)
        rm, rv = sd[prefix + ".running_mean"].to(x.dtype), sd[prefix + ".running_var"].to(x.dtype)
        return (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + BN_EPS) * sd[prefix + ".weight"].view(shape) + sd[prefix + ".bias"].view(shape)
    dims = [0] + list(range(2, x.dim()))
    mean = x.mean(dims)
    var = x.var(dims, unbiased=False)
    shape = [1, -1] + [1] * (x.dim() - 2)
    y = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * sd[prefix + ".weight"].view(shape) + sd[prefix + ".bias"].view(shape)
    if state_out is not None:
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            state_out[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * sd[prefix + ".running_mean"] + BN_MOMENTUM * mean
            state_out[prefix + ".running_var"] = (1 - BN_MOMENTUM) * sd[prefix + ".running_var"] + BN_MOMENTUM * var * n / max(n - 1, 1)
    return y


def _basic_block(x, sd, p, stride, so):
    """torchvision.models.resnet.BasicBlock.forward"""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1", so))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2", so)
    idn = x
    if (p + ".downsample.0.weight") in sd:
        idn = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), sd, p + ".downsample.1", so)
    return F.relu(out + idn)


def encoder_forward(x, sd, so=None, prefix="model.encoder"):
    """smp ResNetEncoder.forward (depth 5): [identity, stem, layer1(maxpool), layer2, layer3, layer4]"""
    feats = [x]
    h = F.relu(_bn(F.conv2d(x, sd[prefix + ".conv1.weight"], None, 2, 3), sd, prefix + ".bn1", so))
    feats.append(h)
    h = F.max_pool2d(h, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        h = _basic_block(h, sd, f"{prefix}.layer{li}.0", stride, so)
        h = _basic_block(h, sd, f"{prefix}.layer{li}.1", 1, so)
        feats.append(h)
    return feats


def decoder_block(x, sd, p, so=None):
    """DecoderBlock.forward, pcrlv2_model.py:113-128"""
    x = F.interpolate(x, scale_factor=2, mode="nearest")                                                        # :114
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.0.weight"], None, 1, 1), sd, p + ".conv1.1", so))               # :119
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv2.0.weight"], None, 1, 1), sd, p + ".conv2.1", so))               # :120
    d = p + ".deep_supervision_head"
    m = F.relu(_bn(F.conv2d(x, sd[d + ".0.weight"], sd[d + ".0.bias"], 1, 1), sd, d + ".1", so))               # :103-105,123
    x_mask = F.conv2d(m, sd[d + ".3.weight"], sd[d + ".3.bias"])                                               # :106
    x_pro = _bn(F.adaptive_avg_pool2d(x, (1, 1)).view(x.shape[0], -1), sd, p + ".bn", so)                      # :125-126
    q = p + ".predictor_head"
    h = F.relu(_bn(F.linear(x_pro, sd[q + ".0.weight"], sd[q + ".0.bias"]), sd, q + ".1", so))                 # :108-110
    x_pre = F.linear(h, sd[q + ".3.weight"], sd[q + ".3.bias"])                                                # :111,127
    return x, x_pro, x_pre, x_mask


def model_forward(x, sd, local=False, so=None, training=True):
    """PCRLv2.forward, pcrlv2_model.py:203-209 (the decoder is always called with local=False there: quirk kept).
    training=False: the same forward under nn.Module.eval() (running statistics, no buffer updates)."""
    global EVAL
    if not training:
        EVAL = True
        try:
            return model_forward(x, sd, local, None, True)
        finally:
            EVAL = False
    feats = encoder_forward(x, sd, so)
    h = feats[1:][::-1][0]                                                                                      # :177-180
    outs, masks_mid = [], []
    for i in range(5):
        h, pro, pre, x_mask = decoder_block(h, sd, f"model.decoder.blocks.{i}", so)
        outs.append((pro, pre))
        masks_mid.append(F.interpolate(x_mask, scale_factor=2 ** (4 - i), mode="bilinear"))                     # :190
    masks = None
    if not local:
        masks = F.conv2d(h, sd["model.segmentation_head.0.weight"], sd["model.segmentation_head.0.bias"], 1, 1)  # :207-208
    return outs, masks, masks_mid


def cos_loss(o1, o2):
    """train_2d.py:111-117"""
    k = random.randint(0, len(o1) - 1)
    s1, s2 = o1[k], o2[k]
    cos = torch.nn.CosineSimilarity()
    return -(cos(s1[1], s2[0].detach()).mean() + cos(s2[1], s1[0].detach()).mean()) * 0.5, k


def step_losses(sd, batch, epoch, so=None):
    """train_2d.py:139-168 -> dict(loss, loss1, loss2, loss4, local_loss, index2)"""
    x1, x2, gt, _gt2, local_views = batch
    n = x1.shape[0]
    o1, mask1, mid1 = model_forward(x1, sd, so=so)
    o2, _mask2, _ = model_forward(x2, sd, so=so)
    loss2, index2 = cos_loss(o1, o2)
    ol, _, _ = model_forward(torch.cat(local_views, dim=0), sd, local=True, so=so)
    ol = [torch.stack(t) for t in ol]
    local_loss = 0.0
    for i in range(len(local_views)):
        tmp = [t[:, n * i: n * (i + 1)] for t in ol]
        local_loss = local_loss + cos_loss(o1, tmp)[0]
        local_loss = local_loss + cos_loss(o2, tmp)[0]
    local_loss = local_loss / (2 * len(local_views))
    loss1 = F.mse_loss(mask1, gt)
    beta = 0.5 * (1.0 + math.cos(math.pi * epoch / BETA_PERIOD))
    loss4 = beta * F.mse_loss(mid1[index2], gt)
    return {"loss": loss1 + loss2 + local_loss + loss4, "loss1": loss1, "loss2": loss2, "loss4": loss4, "local_loss": local_loss,
            "index2": index2, "out1": o1, "mask1": mask1, "mid1": mid1}


def synthetic_batch(b, size, local_size, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(b, 3, size, size, generator=g, dtype=dtype)
    x2 = x1 + 0.1 * torch.randn(b, 3, size, size, generator=g, dtype=dtype)
    gt = torch.rand(b, 3, size, size, generator=g, dtype=dtype)
    locs = [torch.randn(b, 3, local_size, local_size, generator=g, dtype=dtype) for _ in range(6)]
    return x1, x2, gt, gt.clone(), locs


def fill_loss_inputs(b=4, nlocal=6, size=32, seed=31, dtype=torch.float64, channels=(256, 128, 64, 32, 16)):
    """Closed-form stand-ins for what the three forwards of train_2d.py:139-147 hand to the loss assembly (the MODEL is not part of this pin:
    smp / torchvision are absent, 'parity unpinned' -- only train_2d.py:111-117,139-168 is): five scales of [projection, prediction] features for
    view 1 / view 2 ([b, C_k]) and the concatenated local views ([nlocal * b, C_k]), the reconstruction, the five deep-supervision maps and the
    target ([b, 3, size, size]).  View 2 and the local features are correlated with view 1 (cosine terms away from 0)."""
    import sys as _sys
    import os as _os
    _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
    import pcrlv2_oracle as O3
    import numpy as np

    def u(n, sd):
        return torch.from_numpy(np.ascontiguousarray(O3._hash_uniform(n, sd))).to(dtype)
    feats1, feats2, feats_loc = [], [], []
    for k, c in enumerate(channels):
        pro1, pre1 = u(b * c, seed + 10 * k).reshape(b, c), u(b * c, seed + 10 * k + 1).reshape(b, c)
        pro2 = 0.6 * pre1 + 0.4 * u(b * c, seed + 10 * k + 2).reshape(b, c)
        pre2 = 0.6 * pro1 + 0.4 * u(b * c, seed + 10 * k + 3).reshape(b, c)
        prol = 0.5 * pre1.repeat(nlocal, 1) + 0.5 * u(nlocal * b * c, seed + 10 * k + 4).reshape(nlocal * b, c)
        prel = 0.5 * pro2.repeat(nlocal, 1) + 0.5 * u(nlocal * b * c, seed + 10 * k + 5).reshape(nlocal * b, c)
        feats1.append([pro1, pre1])
        feats2.append([pro2, pre2])
        feats_loc.append([prol, prel])
    n = b * 3 * size * size
    target = (0.5 + 0.5 * u(n, seed + 100)).reshape(b, 3, size, size)
    mask1 = (0.5 + 0.4 * u(n, seed + 101)).reshape(b, 3, size, size)
    masks1 = [(0.5 + 0.45 * u(n, seed + 110 + k)).reshape(b, 3, size, size) for k in range(len(channels))]
    return feats1, feats2, feats_loc, mask1, masks1, target
