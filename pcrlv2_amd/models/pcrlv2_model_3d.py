"""PCRLv23d on the MI355X engine -- drop-in for the reference's models/pcrlv2_model_3d.py.

Same classes, constructor arguments, forward signatures, return tuples, attribute names and
state_dict (169 entries); the compute goes through libpcrl_hip.so (hand-written
gfx950 kernels) via `pcrlv2_amd.functions`.  The torch.nn layers instantiated below are PARAMETER
CONTAINERS only: they give the reference's parameter names, shapes and default initialisation (so
the same `torch.manual_seed` yields the same initial weights as the reference); their forward()
is never called.  There is no CPU / eager fallback: calling the model on non-GPU tensors raises.

Not supported (raises NotImplementedError at construction): norm in {'gn','in'} and act in
{'prelu','elu'} -- the reference accepts these strings but never instantiates them ('gn' crashes
in the reference itself); in_channels != 1 and n_class != 1.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import config, functions as Fn, ops
from .._lib import ACT_RELU, ACT_SIGMOID


class _Counted:
    """Lazy `num_batches_tracked` bookkeeping: the counter buffers are bumped on the host and written
    to the device tensors only when somebody looks (state_dict / flush), not once per forward."""

    def _init_counter(self, bns):
        self._bns = bns
        self._pending = 0

    def _count_batch(self):
        self._pending += 1

    def flush_counters(self):
        if self._pending:
            for bn in self._bns:
                bn.num_batches_tracked += self._pending
            self._pending = 0


class LUConv(nn.Module, _Counted):
    """reference: models/pcrlv2_model_3d.py:6-34"""

    def __init__(self, in_chan, out_chan, act, norm):
        super(LUConv, self).__init__()
        self.conv1 = nn.Conv3d(in_chan, out_chan, kernel_size=3, padding=1)
        if norm == 'bn':
            self.bn1 = nn.BatchNorm3d(num_features=out_chan, momentum=0.1, affine=True)
        elif norm in ('gn', 'in'):
            raise NotImplementedError("normalization type {} has no gfx950 kernel (reference default and only "
                                      "working configuration is 'bn')".format(norm))
        else:
            raise ValueError('normalization type {} is not supported'.format(norm))
        if act == 'relu':
            self._act = ACT_RELU
        elif act == 'sigmoid':
            self._act = ACT_SIGMOID
        elif act in ('prelu', 'elu'):
            raise NotImplementedError("activation type {} has no gfx950 kernel".format(act))
        else:
            raise ValueError('activation type {} is not supported'.format(act))
        self.compute_dtype = config.default_compute_dtype()
        self._packed = ops.PackedWeights("conv3")
        self._init_counter([self.bn1])

    def forward(self, x):
        if self.conv1.in_channels == 1:
            x = x.float().contiguous()
        else:
            x = ops.to_act(x, self.compute_dtype)
        return Fn.LUConvFn.apply(x, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias, self)


def _make_nConv(in_channel, depth, act, norm, double_chnnel=False):
    """reference: models/pcrlv2_model_3d.py:37-45"""
    if double_chnnel:
        layer1 = LUConv(in_channel, 32 * (2 ** (depth + 1)), act, norm)
        layer2 = LUConv(32 * (2 ** (depth + 1)), 32 * (2 ** (depth + 1)), act, norm)
    else:
        layer1 = LUConv(in_channel, 32 * (2 ** depth), act, norm)
        layer2 = LUConv(32 * (2 ** depth), 32 * (2 ** depth) * 2, act, norm)
    return nn.Sequential(layer1, layer2)


class UpTransition(nn.Module, _Counted):
    """reference: models/pcrlv2_model_3d.py:48-72"""

    def __init__(self, inChans, outChans, depth, act, norm):
        super(UpTransition, self).__init__()
        self.depth = depth
        self.up_conv = nn.ConvTranspose3d(inChans, outChans, kernel_size=2, stride=2)
        self.ops = _make_nConv(outChans, depth, act, norm, double_chnnel=True)
        channels = 32 * (2 ** depth) * 2
        self.bn = nn.BatchNorm1d(channels)
        self.predictor_head = nn.Sequential(nn.Linear(channels, 2 * channels),
                                            nn.BatchNorm1d(2 * channels),
                                            nn.ReLU(inplace=True),
                                            nn.Linear(2 * channels, channels))
        self.deep_supervision_head = LUConv(channels, 1, 'sigmoid', norm)
        if act != 'relu':
            raise NotImplementedError("UpTransition is implemented for act='relu' (the reference default)")
        self.compute_dtype = config.default_compute_dtype()
        self._packed_up = ops.PackedWeights("convt")
        self._init_counter([self.bn, self.predictor_head[1]])

    def _count_batch_heads(self):
        self._count_batch()

    def forward(self, x):
        l0, l1, ld, ph = self.ops[0], self.ops[1], self.deep_supervision_head, self.predictor_head
        return Fn.UpStageFn.apply(
            x, self.up_conv.weight, self.up_conv.bias,
            l0.conv1.weight, l0.conv1.bias, l0.bn1.weight, l0.bn1.bias,
            l1.conv1.weight, l1.conv1.bias, l1.bn1.weight, l1.bn1.bias,
            self.bn.weight, self.bn.bias, ph[0].weight, ph[0].bias, ph[1].weight, ph[1].bias, ph[3].weight, ph[3].bias,
            ld.conv1.weight, ld.conv1.bias, ld.bn1.weight, ld.bn1.bias, self)


class OutputTransition(nn.Module):
    """reference: models/pcrlv2_model_3d.py:75-83"""

    def __init__(self, inChans, n_labels):
        super(OutputTransition, self).__init__()
        self.final_conv = nn.Conv3d(inChans, n_labels, kernel_size=1)
        self.sigmoid = nn.Sigmoid()
        if n_labels != 1:
            raise NotImplementedError("n_class != 1 has no gfx950 kernel (the pre-training path uses n_class=1)")
        self.compute_dtype = config.default_compute_dtype()

    def forward(self, x):
        return Fn.OutFn.apply(x, self.final_conv.weight, self.final_conv.bias, self)


class DownTransition(nn.Module):
    """reference: models/pcrlv2_model_3d.py:86-92"""

    def __init__(self, in_channel, depth, act, norm):
        super(DownTransition, self).__init__()
        self.ops = _make_nConv(in_channel, depth, act, norm)

    def forward(self, x):
        return self.ops(x)


class _MaxPool3d2(nn.MaxPool3d):
    """`self.maxpool` of the reference (:100); forward goes to the gfx950 kernel."""

    def __init__(self):
        super().__init__(2)
        self.compute_dtype = config.default_compute_dtype()

    def forward(self, x):
        dt = self.compute_dtype
        return Fn.MaxPoolFn.apply(ops.to_act(x, dt), dt)


class PCRLv23d(nn.Module):
    """reference: models/pcrlv2_model_3d.py:95-133"""

    def __init__(self, n_class=1, act='relu', norm='bn', in_channels=1, low_dim=128, student=False):
        super(PCRLv23d, self).__init__()
        if in_channels != 1:
            raise NotImplementedError("in_channels != 1 has no gfx950 first-layer kernel (LUNA volumes are 1-channel)")
        self.compute_dtype = config.default_compute_dtype()
        self.maxpool = _MaxPool3d2()
        self.down_tr64 = DownTransition(in_channels, 0, act, norm)
        self.down_tr128 = DownTransition(64, 1, act, norm)
        self.down_tr256 = DownTransition(128, 2, act, norm)
        self.down_tr512 = DownTransition(256, 3, act, norm)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))  # unused, kept like the reference (:105)
        self.up_tr256 = UpTransition(512, 512, 2, act, norm)
        self.up_tr128 = UpTransition(256, 256, 1, act, norm)
        self.up_tr64 = UpTransition(128, 128, 0, act, norm)
        self.out_tr = OutputTransition(64, n_class)
        self.sigmoid = nn.Sigmoid()                       # unused, kept like the reference (:110)

    # ---- engine controls (not in the reference) ----
    def set_compute_dtype(self, dt):
        """float32 (exact parity mode) or bfloat16 (MFMA throughput mode) for activations / packed weights."""
        if isinstance(dt, str):
            dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[dt]
        if dt not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dt
        return self

    def flush_counters(self):
        for m in self.modules():
            if isinstance(m, _Counted):
                m.flush_counters()

    def _stage_modules(self):
        if not hasattr(self, "_stages"):
            object.__setattr__(self, "_stages", [m for m in self.modules() if isinstance(m, (LUConv, UpTransition, OutputTransition))])
        return self._stages

    def state_dict(self, *args, **kwargs):
        self.flush_counters()
        return super().state_dict(*args, **kwargs)

    def forward(self, x, local=False):
        if not self.training:
            raise NotImplementedError("PCRLv23d on the MI355X engine implements the pre-training (train-mode) path only")
        if not x.is_cuda:
            raise RuntimeError("PCRLv23d (pcrlv2_amd) runs on the GPU only: input is on %s and there is no CPU fallback" % x.device)
        b = x.shape[0]
        pass_idx = ops.next_pass()          # 0 = first forward since the last optimizer step (its backward runs last)
        for m in self._stage_modules():
            m._pass_idx = pass_idx
        self.skip_out64 = self.down_tr64(x)
        self.skip_out128 = self.down_tr128(self.maxpool(self.skip_out64))
        self.skip_out256 = self.down_tr256(self.maxpool(self.skip_out128))
        self.out512 = self.down_tr512(self.maxpool(self.skip_out256))
        middle_masks = []
        middle_features = []
        out_up_256, pro_256, pre_256, middle_masks_256 = self.up_tr256(self.out512)
        out_up_128, pro_128, pre_128, middle_masks_128 = self.up_tr128(out_up_256)
        out_up_64, pro_64, pre_64, middle_masks_64 = self.up_tr64(out_up_128)
        if not local:
            middle_masks.append(Fn.TrilinearFn.apply(middle_masks_256, 4))
            middle_masks.append(Fn.TrilinearFn.apply(middle_masks_128, 2))
            middle_masks.append(middle_masks_64)
        middle_features.append([pro_256, pre_256])
        middle_features.append([pro_128, pre_128])
        middle_features.append([pro_64, pre_64])
        out = self.out_tr(out_up_64)
        return out, middle_features, middle_masks
