"""PCRLv2 (2D, ResNet-18 U-Net) on the MI355X engine -- drop-in for the reference's models/pcrlv2_model.py  (SURVEY 8f N1).

Same public classes / forward signatures / return tuples; the module tree reproduces the attribute paths the reference gets
from `smp.Unet('resnet18', in_channels=3, classes=n_class)` with its decoder replaced by `PCRLv2Decoder`
(pcrlv2_model.py:197-209): `model.encoder` (torchvision ResNet-18 without `fc`: conv1, bn1, layer{1..4}.{0,1}.{conv1,bn1,conv2,
bn2}, layer{2..4}.0.downsample.{0,1}), `model.decoder.blocks[i]` (conv1 / conv2 = Sequential(conv, bn, relu), bn, predictor_head,
deep_supervision_head) and `model.segmentation_head` (Sequential(conv3x3, Identity, Identity)) -- so
`model.model.encoder.state_dict()` (what train_2d.py:99 saves) has the torchvision key names README.md:40-44 loads.

segmentation_models_pytorch and torchvision are NOT dependencies: the two public definitions this file needs (ResNet-18
BasicBlock encoder, smp Conv2dReLU = Conv2d(bias=False) + BatchNorm2d + ReLU) are restated.  PARITY UNPINNED for the encoder
(the reference cannot be imported here; no golden vectors exist) -- the decoder follows pcrlv2_model.py:68-194 line
by line and is checked against a plain-PyTorch restatement in tests/.

torch.nn layers are parameter containers (names, shapes, initialisers); the arithmetic runs in libpcrl_hip.so through
`pcrlv2_amd.functions2d`.  Difference on purpose: `encoder_weights` -- the reference's smp default downloads ImageNet weights at
construction (impossible offline); here the encoder is randomly initialised (torchvision's scheme) unless `encoder_weights` is a
path to a state_dict file.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import config, functions2d as Fn2, ops, ops2d
from .._lib import ACT_NONE, ACT_RELU
from .pcrlv2_model_3d import _Counted


class _Unit(_Counted):
    """Engine-side record of one conv (+BatchNorm2d +activation) unit: geometry, packed weights, counters.  Not an nn.Module --
    the parameters stay where the reference's module tree has them."""

    def __init__(self, conv: nn.Conv2d, bn, act, up=False):
        self.conv, self.bn_module = conv, bn
        self.stride, self.pad, self.up, self.act = conv.stride[0], conv.padding[0], int(up), act
        self.compute_dtype = config.default_compute_dtype()
        self._packed = ops2d.PackedConv2d()
        self._pass_idx = 1
        self._init_counter([bn] if bn is not None else [])

    def __call__(self, x):
        c, n = self.conv, self.bn_module
        if n is None:
            return Fn2.ConvFn.apply(x, c.weight, c.bias, self)
        return Fn2.ConvBNActFn.apply(x, c.weight, c.bias, n.weight, n.bias, self)


def _bn2d(c):
    return nn.BatchNorm2d(c, momentum=ops.BN_MOMENTUM)


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock (expansion 1)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = _bn2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = _bn2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), _bn2d(planes))
        self._u1 = _Unit(self.conv1, self.bn1, ACT_RELU)
        self._u2 = _Unit(self.conv2, self.bn2, ACT_NONE)
        self._ud = _Unit(self.downsample[0], self.downsample[1], ACT_NONE) if self.downsample is not None else None

    def _units(self):
        return [u for u in (self._u1, self._u2, self._ud) if u is not None]

    def forward(self, x):
        t = self._u2(self._u1(x))
        identity = x if self._ud is None else self._ud(x)
        return Fn2.AddReluFn.apply(t, identity, self._u1.compute_dtype)


class ResNetEncoder(nn.Module):
    """smp.encoders.resnet.ResNetEncoder('resnet18', depth=5, in_channels=3): torchvision ResNet-18 minus `fc`;
    forward returns the six feature maps [x, stem, layer1, layer2, layer3, layer4]."""

    out_channels = (3, 64, 64, 128, 256, 512)

    def __init__(self, in_channels=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = _bn2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))
        self._stem = _Unit(self.conv1, self.bn1, ACT_RELU)
        for m in self.modules():   # torchvision.models.resnet.ResNet.__init__
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _units(self):
        us = [self._stem]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                us += blk._units()
        return us

    def forward(self, x):
        dt = self._stem.compute_dtype
        feats = [x]
        h = self._stem(ops2d.image_to_act(x, dt, 8))
        feats.append(h)
        h = Fn2.MaxPool2dFn.apply(h, dt)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            h = layer(h)
            feats.append(h)
        return feats

    def forward_last(self, x):
        """The last feature map only -- all the decoder reads (pcrlv2_model.py:115-117 ignores the skips) -- through ONE autograd node
        (functions2d.EncoderFn); with PCRL_FUSED_ENCODER_2D=0 the per-unit path above (same values, bit for bit)."""
        if not Fn2.FUSED_ENCODER or x.dtype != torch.float32:
            return self.forward(x)[-1]
        params = []
        for u in self._units():
            params += [u.conv.weight, u.bn_module.weight, u.bn_module.bias]
        return Fn2.EncoderFn.apply(x, self, *params)


def initialize_decoder(module):
    """reference pcrlv2_model.py:23-38"""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


def initialize_head(module):
    """reference pcrlv2_model.py:41-46"""
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


def _conv2d_relu(cin, cout):
    """smp.base.modules.Conv2dReLU(use_batchnorm=True): Sequential(Conv2d(bias=False), BatchNorm2d, ReLU)"""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), _bn2d(cout), nn.ReLU(inplace=True))


class _Attention(nn.Module):
    """smp.base.modules.Attention(None): identity"""

    def __init__(self):
        super().__init__()
        self.attention = nn.Identity()

    def forward(self, x):
        return x


class DecoderBlock(nn.Module, _Counted):
    """reference pcrlv2_model.py:68-128.  The skip input is ignored there too (the concatenation is commented out, :115-117)."""

    def __init__(self, in_channels, skip_channels, out_channels, use_batchnorm=True, attention_type=None):
        super().__init__()
        if not use_batchnorm or attention_type is not None:
            raise NotImplementedError("DecoderBlock: only use_batchnorm=True, attention_type=None (the reference's setting) have gfx950 kernels")
        self.conv1 = _conv2d_relu(in_channels, out_channels)
        self.attention1 = _Attention()
        self.conv2 = _conv2d_relu(out_channels, out_channels)
        self.attention2 = _Attention()
        self.bn = nn.BatchNorm1d(out_channels, momentum=ops.BN_MOMENTUM)
        self.deep_supervision_head = nn.Sequential(nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1), _bn2d(out_channels),
                                                   nn.ReLU(inplace=True), nn.Conv2d(out_channels, 3, kernel_size=1))
        self.predictor_head = nn.Sequential(nn.Linear(out_channels, 2 * out_channels),
                                            nn.BatchNorm1d(2 * out_channels, momentum=ops.BN_MOMENTUM), nn.ReLU(inplace=True),
                                            nn.Linear(2 * out_channels, out_channels))
        self.compute_dtype = config.default_compute_dtype()
        self._pass_idx = 1
        self._u1 = _Unit(self.conv1[0], self.conv1[1], ACT_RELU, up=True)     # F.interpolate(nearest x2) fused into conv1's gather (:114)
        self._u2 = _Unit(self.conv2[0], self.conv2[1], ACT_RELU)
        ds = self.deep_supervision_head
        self._ud0 = _Unit(ds[0], ds[1], ACT_RELU)
        self._ud3 = _Unit(ds[3], None, ACT_NONE)
        self._init_counter([self.bn, self.predictor_head[1]])

    def _count_batch_heads(self):
        self._count_batch()

    def _units(self):
        return [self._u1, self._u2, self._ud0, self._ud3]

    def forward(self, x, skip=None, want_mask=True):
        """-> (x, x_pro, x_pre, x_mask).  `want_mask` (engine hint, not in the reference): False leaves x_mask None -- the head's
        convolution and BatchNorm2d statistics still run (the module's state is the reference's), see functions2d.DecoderBlockFn."""
        c1, c2, ds, ph = self.conv1, self.conv2, self.deep_supervision_head, self.predictor_head
        return Fn2.DecoderBlockFn.apply(x, c1[0].weight, c1[1].weight, c1[1].bias, c2[0].weight, c2[1].weight, c2[1].bias,
                                        ds[0].weight, ds[0].bias, ds[1].weight, ds[1].bias, ds[3].weight, ds[3].bias,
                                        self.bn.weight, self.bn.bias, ph[0].weight, ph[0].bias, ph[1].weight, ph[1].bias,
                                        ph[3].weight, ph[3].bias, self, bool(want_mask))


class PCRLv2Decoder(nn.Module):
    """reference pcrlv2_model.py:131-194"""

    def __init__(self, encoder_channels=512, n_class=3, decoder_channels=(256, 128, 64, 32, 16), n_blocks=5, use_batchnorm=True,
                 center=False, attention_type=None):
        super().__init__()
        if n_blocks != len(decoder_channels):
            raise ValueError("Model depth is {}, but you provide `decoder_channels` for {} blocks.".format(n_blocks, len(decoder_channels)))
        if center:
            raise NotImplementedError("PCRLv2Decoder(center=True) is never used by the reference and has no gfx950 path")
        encoder_channels = encoder_channels[1:][::-1]
        head_channels = encoder_channels[0]
        in_channels = [head_channels] + list(decoder_channels[:-1])
        skip_channels = list(encoder_channels[1:]) + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList([DecoderBlock(i, s, o, use_batchnorm=use_batchnorm, attention_type=attention_type)
                                     for i, s, o in zip(in_channels, skip_channels, decoder_channels)])
        initialize_decoder(self.blocks)

    def forward(self, features, local=False, _mask_scales=None, _upsample=True):
        # NOTE (reference quirk, kept): PCRLv2.forward never passes `local`, so the deep-supervision maps are upsampled for the
        # local views too (pcrlv2_model.py:205) -- `local=True` only skips the segmentation head.
        # Engine hints (not in the reference; PCRLv2.forward_engine): `_mask_scales` -- the block indices whose deep-supervision map is
        # wanted (None: all five); `_upsample` False returns the maps at their own resolution (None where not wanted).
        features = features[1:][::-1]
        x = self.center(features[0])
        decoder_outs, middle_masks = [], []
        for i, block in enumerate(self.blocks):
            want = _mask_scales is None or i in _mask_scales
            x, x_pro, x_pre, x_mask = block(x, None, want_mask=want)
            decoder_outs.append((x_pro, x_pre))
            if not _upsample:
                middle_masks.append(x_mask)
            elif not local:
                middle_masks.append(Fn2.BilinearFn.apply(x_mask, 2 ** (4 - i)) if x_mask is not None else None)
        return decoder_outs, x, middle_masks


class _SegmentationModel(nn.Module):
    """Shape of smp.Unet as the reference uses it: .encoder, .decoder, .segmentation_head (.classification_head = None)."""

    def __init__(self, n_class, encoder_weights=None):
        super().__init__()
        self.encoder = ResNetEncoder(3)
        self.decoder = PCRLv2Decoder(self.encoder.out_channels)
        self.segmentation_head = nn.Sequential(nn.Conv2d(16, n_class, kernel_size=3, padding=1), nn.Identity(), nn.Identity())
        self.classification_head = None
        initialize_head(self.segmentation_head)
        self.name = "u-resnet18"
        if encoder_weights is not None:
            sd = torch.load(encoder_weights, map_location="cpu")
            sd = sd.get("state_dict", sd)
            sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
            self.encoder.load_state_dict(sd)


class PCRLv2(nn.Module):
    """reference pcrlv2_model.py:197-209"""

    def __init__(self, n_class=3, low_dim=128, encoder_weights=None):
        super().__init__()
        self.model = _SegmentationModel(n_class, encoder_weights)
        self.compute_dtype = config.default_compute_dtype()
        self._seg = _Unit(self.model.segmentation_head[0], None, ACT_NONE)

    # ---- engine controls (not in the reference) ----
    def _all_units(self):
        us = self.model.encoder._units()
        for b in self.model.decoder.blocks:
            us += b._units()
        return us + [self._seg]

    def set_compute_dtype(self, dt):
        dt = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(dt, dt) if isinstance(dt, str) else dt
        if dt not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dt
        for u in self._all_units():
            u.compute_dtype = dt
        for b in self.model.decoder.blocks:
            b.compute_dtype = dt
        return self

    def flush_counters(self):
        for u in self._all_units():
            u.flush_counters()
        for b in self.model.decoder.blocks:
            b.flush_counters()

    def state_dict(self, *args, **kwargs):
        self.flush_counters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        for u in self._all_units():
            u._pending = 0
        for b in self.model.decoder.blocks:
            b._pending = 0
        out = super().load_state_dict(state_dict, *args, **kwargs)
        ops.bump_weights_epoch()
        return out

    # ---- model.eval(): what a consumer of the saved encoder / model runs (README.md:31-45) -- the same convolution / normalisation kernels on
    #      the RUNNING statistics, nothing updated, no autograd graph (the 2D model raised outside train mode) ----
    @staticmethod
    def _unit_eval(u, x, dt, out_f32=False):
        c, n = u.conv, u.bn_module
        if n is None:
            return ops2d.conv2d_forward(x, c.weight, c.bias, u._packed, u.stride, u.pad, 0, dt, want_stats=False, out_f32=True)[0]
        y = ops2d.conv2d_forward(x, c.weight, c.bias, u._packed, u.stride, u.pad, u.up, dt, want_stats=False)[0]
        N, H, W, C = ops2d.dims2(y)
        scale, shift = ops.bn_eval_coef(n.weight, n.bias, n.running_mean, n.running_var)
        return ops.bn_act_apply(y, scale, shift, N * H * W, C, u.act, dt)

    @torch.no_grad()
    def _forward_eval(self, x, local=False):
        if not x.is_cuda:
            raise RuntimeError("PCRLv2 (pcrlv2_amd) runs on the GPU only: input is on %s and there is no CPU fallback" % x.device)
        dt, ue = self.compute_dtype, self._unit_eval
        enc = self.model.encoder
        h = ue(enc._stem, ops2d.image_to_act(x.float(), dt, 8), dt)
        h = ops2d.maxpool_forward(h, dt)[0]
        for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
            for blk in layer:
                t = ue(blk._u2, ue(blk._u1, h, dt), dt)
                idn = h if blk._ud is None else ue(blk._ud, h, dt)
                h = ops2d.add_relu_forward(t, ops2d.to_act2(idn, dt), dt)
        decoder_outputs, middle_masks = [], []
        for i, blk in enumerate(self.model.decoder.blocks):
            h = ue(blk._u2, ue(blk._u1, h, dt), dt)
            ph = blk.predictor_head
            g = ops2d.gap_forward(h, dt)
            x_pro = ops.bn1d_eval(g, blk.bn.weight, blk.bn.bias, blk.bn.running_mean, blk.bn.running_var, relu=False)
            hid = ops.bn1d_eval(ops.linear_forward(x_pro, ph[0].weight, ph[0].bias), ph[1].weight, ph[1].bias, ph[1].running_mean, ph[1].running_var, relu=True)
            x_pre = ops.linear_forward(hid, ph[3].weight, ph[3].bias)
            decoder_outputs.append((x_pro, x_pre))
            x_mask = ue(blk._ud3, ue(blk._ud0, h, dt), dt)
            # (reference quirk, kept: PCRLv2.forward never hands `local` to the decoder -- the maps are upsampled for the local views too)
            middle_masks.append(ops2d.bilinear_forward(ops2d.to_act2(x_mask, torch.float32), 2 ** (4 - i)))
        masks = None if local else ue(self._seg, h, dt)
        return decoder_outputs, masks, middle_masks

    def _begin_pass(self, x):
        if not x.is_cuda:
            raise RuntimeError("PCRLv2 (pcrlv2_amd) runs on the GPU only: input is on %s and there is no CPU fallback" % x.device)
        pass_idx = ops.next_pass()
        for u in self._all_units():
            u._pass_idx = pass_idx
        for b in self.model.decoder.blocks:
            b._pass_idx = pass_idx

    def forward(self, x, local=False):
        """-> ([(pro, pre) x 5], masks [b,n_class,H,W] | None, [mask x 5])"""
        if not self.training:
            return self._forward_eval(x, local)
        self._begin_pass(x)
        features = [None] * 5 + [self.model.encoder.forward_last(x)]      # the decoder reads the last feature map only
        decoder_outputs, h, middle_masks = self.model.decoder(features)
        masks = None
        if not local:
            masks = self._seg(h)
        return decoder_outputs, masks, middle_masks

    def forward_engine(self, x, mask_scale=None):
        """The training step's form of forward (train_2d.step_losses; not in the reference): everything with STATE runs exactly as in
        forward() -- every convolution in front of a BatchNorm, every BatchNorm1d of the heads -- but what has neither state nor a consumer
        in train_2d.py:139-168 is not computed: the segmentation head (its loss is taken by functions2d.SegMSEFn from the returned decoder
        output), the bilinear upsampling, and the deep-supervision maps of every scale but `mask_scale` (the scale the first cos_loss
        draws; None: no map at all -- the second view and the local views, whose maps the reference computes and never reads).
        -> ([(pro, pre) x 5], decoder output (activation), deep-supervision map of `mask_scale` at its own resolution | None)"""
        if not self.training:
            raise RuntimeError("PCRLv2.forward_engine is the TRAINING step's forward; in eval mode call the model (forward)")
        self._begin_pass(x)
        features = [None] * 5 + [self.model.encoder.forward_last(x)]
        decoder_outputs, h, low = self.model.decoder(features, _mask_scales=() if mask_scale is None else (mask_scale,), _upsample=False)
        return decoder_outputs, h, (low[mask_scale] if mask_scale is not None else None)
