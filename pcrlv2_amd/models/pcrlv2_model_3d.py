"""PCRLv23d on the MI355X engine -- drop-in for the reference's models/pcrlv2_model_3d.py.

Same public classes, constructor arguments, forward signatures, return tuples, attribute names and state_dict
(169 entries in the same order); the arithmetic runs in libpcrl_hip.so (hand-written gfx950 kernels)
through `pcrlv2_amd.functions`.  torch.nn layers appear here ONLY as parameter containers: they provide the reference's
parameter names / shapes / default initialisation (so an identical `torch.manual_seed` yields identical initial
weights); their forward() is never called.  There is no CPU / eager fallback: non-GPU inputs raise.
`model.eval()` runs the same kernels on the running statistics (inference; `_forward_eval`).

Constructor variants the reference accepts but train_3d.py:45 never instantiates -- act in {'elu','prelu'}, norm='in',
in_channels != 1, n_class != 1 (models/pcrlv2_model_3d.py:15-16,22-25,98) -- run on the same library through general-purpose routes
(ELU as an activation code of the fused BatchNorm kernels; PReLU as a streaming pass behind the normalisation; InstanceNorm3d as
GroupNorm with one channel per group; the first layer zero-padded to 32 input channels; one C -> 1 pass per output class) and are
pinned against the real reference built with the same arguments (tests/golden/v_*.npz, tests/test_variants_gpu.py).  They are not
tuned: the pre-training path is PCRLv23d() with its defaults.  norm='gn' crashes in the reference; here it is the optional
GroupNorm + SiLU mode described below.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import config, functions as Fn, ops
from .._lib import ACT_ELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU

# 'silu' and norm='gn' are OPTIONAL, NON-REFERENCE modes (BASELINE.json's north_star names GroupNorm + SiLU; the reference rejects
# 'silu' and crashes on 'gn'): PCRLv23d(norm='gn', act='silu') runs conv -> GroupNorm(8) -> SiLU in every LUConv except
# the 1-channel deep-supervision heads, which keep BatchNorm + sigmoid (GroupNorm(8, 1) cannot exist).  Default = the reference.
_ACTS = {"relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "silu": ACT_SILU, "elu": ACT_ELU, "prelu": ACT_NONE}   # prelu: a pass of its own behind the normalisation


class _Counted:
    """`num_batches_tracked` bookkeeping done lazily: bumped on the host per forward, written to the device buffers
    only when somebody looks (state_dict / flush_counters)."""

    def _init_counter(self, bns):
        self._bns, self._pending = bns, 0

    def _count_batch(self):
        self._pending += 1

    def flush_counters(self):
        if self._pending:
            for bn in self._bns:
                bn.num_batches_tracked += self._pending
            self._pending = 0


class LUConv(nn.Module, _Counted):
    """conv3x3x3(pad 1, bias) -> BatchNorm3d(batch statistics) -> activation      [reference :6-34]"""

    def __init__(self, in_chan, out_chan, act, norm):
        super().__init__()
        if norm not in ("bn", "gn", "in"):
            raise ValueError('normalization type {} is not supported'.format(norm))
        if act not in _ACTS or (act == "silu" and norm != "gn"):    # the reference rejects 'silu' (:30); only the optional 'gn' mode takes it
            raise ValueError('activation type {} is not supported'.format(act))
        self.conv1 = nn.Conv3d(in_chan, out_chan, 3, padding=1)               # container: weight [Co,Ci,3,3,3], bias [Co]
        self._inorm = norm == "in"
        # per-sample statistics pooled over channel groups: GroupNorm(8) (optional mode) or one channel per group = InstanceNorm3d (:15-16);
        # the 1-channel heads keep BatchNorm in 'gn' mode (GroupNorm(8, 1) cannot exist) and take ops.luconv_forward's InstanceNorm route in 'in' mode
        self._gn_groups = (8 if norm == "gn" else out_chan) if (norm in ("gn", "in") and out_chan > 1) else 0
        if norm == "in":
            self.bn1 = nn.InstanceNorm3d(out_chan, momentum=ops.BN_MOMENTUM, affine=True)   # container: affine only (no running statistics, :16)
        elif self._gn_groups:
            self.bn1 = nn.GroupNorm(8, out_chan)                              # the reference's attribute name for either norm (:13-14)
        else:
            self.bn1 = nn.BatchNorm3d(out_chan, momentum=ops.BN_MOMENTUM)      # container: affine + running statistics
        self._prelu = act == "prelu"
        if self._prelu:
            self.activation = nn.PReLU(out_chan)                              # container: the slope vector, registered after bn1 like the reference (:23)
        # first layer of a multi-channel model (in_channels != 1, :98): zero-padded to the implicit-GEMM kernels' 32-channel granule
        self._ci_pad = 32 * ((in_chan + 31) // 32) if (in_chan != 1 and in_chan % 32) else 0
        self._act = _ACTS[act]
        self.compute_dtype = config.default_compute_dtype()
        self._packed = ops.PackedWeights("conv3")
        self._init_counter([] if (self._gn_groups or self._inorm) else [self.bn1])

    def forward(self, x):
        if self._ci_pad:
            x = x.float()
        else:
            x = x.float().contiguous() if self.conv1.in_channels == 1 else ops.to_act(x, self.compute_dtype)
        c, n = self.conv1, self.bn1
        return Fn.LUConvFn.apply(x, c.weight, c.bias, n.weight, n.bias, self)

    def forward_pooled(self, x, pool_only=False):
        """-> (act(bn1(conv1(x))), MaxPool3d(2) of it) as one autograd node (Fn.LUConvPoolFn): what PCRLv23d.forward asks of the second
        LUConv of an encoder stage.  BatchNorm layers with more than one input channel only.  pool_only (the engine's training step): the unpooled
        activation is not stored -- (a _LazySkip that builds it from the saved pre-normalisation tensor on demand, the pooled tensor)."""
        c, n = self.conv1, self.bn1
        out = Fn.LUConvPoolFn.apply(ops.to_act(x, self.compute_dtype), c.weight, c.bias, n.weight, n.bias, self, pool_only)
        if isinstance(out, tuple):
            return out
        return _LazySkip(self._last_saved, self.compute_dtype), out


class _LazySkip:
    """The unpooled output of an encoder stage when the training step did not store it: `PCRLv23d.skip_out64` (etc.) hands out
    act(scale * y + shift) computed from the stage's saved pre-normalisation tensor the first time somebody reads the attribute -- the values the
    stored tensor would have held, detached (the reference's attribute is a graph node; nothing in train_3d.py reads it)."""

    def __init__(self, sv, dt):
        self.sv, self.dt, self.value = sv, dt, None

    def materialize(self):
        if self.value is None:
            sv = self.sv
            N, D, H, W, _, Co = sv.geom
            with torch.no_grad():
                self.value = ops.bn_act_apply(sv.y, sv.scale, sv.shift, N * D * H * W, Co, sv.act, self.dt)
            self.sv = None
        return self.value


def _make_nConv(in_channel, depth, act, norm, double_chnnel=False):
    """Two LUConvs.  Encoder stage d: in -> 32*2^d -> 64*2^d; decoder stage (double_chnnel): in -> 64*2^d -> 64*2^d   [:37-45]"""
    wide = 64 << depth
    mid = wide if double_chnnel else wide // 2
    first, second = LUConv(in_channel, mid, act, norm), LUConv(mid, wide, act, norm)
    # engine hint, not a submodule (no state_dict entry): `first`'s activation has `second` as its only consumer, so `second`'s data gradient may take
    # the first pass of `first`'s BatchNorm backward from its own output tiles (functions.LUConvFn / ops.luconv_backward `bnred`)
    object.__setattr__(second, "_below", first)
    return nn.Sequential(first, second)


class UpTransition(nn.Module, _Counted):
    """ConvTranspose3d(k2,s2) -> two LUConvs -> {global-average projection head, predictor MLP, deep-supervision map}   [:48-72]"""

    def __init__(self, inChans, outChans, depth, act, norm):
        super().__init__()
        c = 64 << depth
        self.depth = depth
        self.up_conv = nn.ConvTranspose3d(inChans, outChans, 2, stride=2)
        self.ops = _make_nConv(outChans, depth, act, norm, double_chnnel=True)
        self.bn = nn.BatchNorm1d(c)
        self.predictor_head = nn.Sequential(nn.Linear(c, 2 * c), nn.BatchNorm1d(2 * c), nn.ReLU(inplace=True), nn.Linear(2 * c, c))
        self.deep_supervision_head = LUConv(c, 1, "sigmoid", norm)
        self._act = _ACTS[act]
        self.compute_dtype = config.default_compute_dtype()
        self._packed_up = ops.PackedWeights("convt")
        self._composed_up = ops.ComposedUpConv()
        self._init_counter([self.bn, self.predictor_head[1]])

    def _count_batch_heads(self):
        self._count_batch()

    def _stage_params(self):
        lu = lambda m: (m.conv1.weight, m.conv1.bias, m.bn1.weight, m.bn1.bias)
        ph = self.predictor_head
        return ((self.up_conv.weight, self.up_conv.bias) + lu(self.ops[0]) + lu(self.ops[1]) + (self.bn.weight, self.bn.bias)
                + (ph[0].weight, ph[0].bias, ph[1].weight, ph[1].bias, ph[3].weight, ph[3].bias) + lu(self.deep_supervision_head))

    def forward(self, x):
        """-> (x, x_pro, x_pre, x_mask)"""
        return Fn.UpStageFn.apply(x, *self._stage_params(), self)


class OutputTransition(nn.Module):
    """sigmoid(conv1x1x1)   [:75-83]"""

    def __init__(self, inChans, n_labels):
        super().__init__()
        self.final_conv = nn.Conv3d(inChans, n_labels, 1)
        self.sigmoid = nn.Sigmoid()   # kept for attribute parity; the sigmoid is fused into the kernel path
        self.compute_dtype = config.default_compute_dtype()

    def forward(self, x):
        return Fn.OutFn.apply(x, self.final_conv.weight, self.final_conv.bias, self)


class DownTransition(nn.Module):
    """encoder stage   [:86-92]"""

    def __init__(self, in_channel, depth, act, norm):
        super().__init__()
        self.ops = _make_nConv(in_channel, depth, act, norm)

    def forward(self, x):
        return self.ops(x)


class _MaxPool3d2(nn.MaxPool3d):
    """`self.maxpool` of the reference (:100) routed to the gfx950 kernel."""

    def __init__(self):
        super().__init__(2)
        self.compute_dtype = config.default_compute_dtype()

    def forward(self, x):
        return Fn.MaxPoolFn.apply(ops.to_act(x, self.compute_dtype), self.compute_dtype)


_ENCODER = (("down_tr64", None, 0), ("down_tr128", 64, 1), ("down_tr256", 128, 2), ("down_tr512", 256, 3))   # name, Cin, depth
_DECODER = (("up_tr256", 512, 2), ("up_tr128", 256, 1), ("up_tr64", 128, 0))                                  # name, C, depth
_SKIPS = ("skip_out64", "skip_out128", "skip_out256", "out512")                                               # attributes stashed by forward (:114-117)
_UPSAMPLE = (4, 2, 1)                                                                                         # trilinear factors of the three masks (:125-127)


class PCRLv23d(nn.Module):
    """reference: models/pcrlv2_model_3d.py:95-133"""

    def __init__(self, n_class=1, act='relu', norm='bn', in_channels=1, low_dim=128, student=False):
        super().__init__()
        self.compute_dtype = config.default_compute_dtype()
        # registration order == the reference's, so state_dict() enumerates the same 169 keys in the same order
        self.maxpool = _MaxPool3d2()
        for name, cin, depth in _ENCODER:
            setattr(self, name, DownTransition(in_channels if cin is None else cin, depth, act, norm))
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))     # unused in the reference too (:105)
        for name, c, depth in _DECODER:
            setattr(self, name, UpTransition(c, c, depth, act, norm))
        self.out_tr = OutputTransition(64, n_class)
        self.sigmoid = nn.Sigmoid()                         # unused in the reference too (:110)

    # ---- engine controls (not in the reference) ----
    def set_compute_dtype(self, dt):
        """float32 (exact parity mode) or bfloat16 (MFMA throughput mode) for activations / packed weights."""
        dt = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(dt, dt) if isinstance(dt, str) else dt
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

    def load_state_dict(self, state_dict, *args, **kwargs):
        """In-place copy like nn.Module's (parameters may live in FusedSGD's flat arena), plus the two things the engine caches:
        pending `num_batches_tracked` bumps are dropped (the loaded counters win) and packed weights are invalidated."""
        for m in self.modules():
            if isinstance(m, _Counted):
                m._pending = 0
        out = super().load_state_dict(state_dict, *args, **kwargs)
        ops.bump_weights_epoch()
        return out

    @torch.no_grad()
    def _forward_eval(self, x, local):
        """`model.eval()` forward (what a consumer of the checkpoint runs for validation, README.md:48-55): the same kernels with every
        BatchNorm on its RUNNING statistics, nothing updated, no autograd graph (inference only: fine-tuning runs in train mode)."""
        dt = self.compute_dtype

        def lu(m, h):
            c, n, gn = m.conv1, m.bn1, m._gn_groups
            w = c.weight
            if m._ci_pad:
                h, w = ops.pad_first_layer(h, w, m._ci_pad, dt)
            return ops.luconv_forward(h, w, c.bias, n.weight, n.bias, None if gn else n.running_mean, None if gn else n.running_var,
                                      m._packed, m._act, dt, training=False, gn_groups=gn, prelu=Fn._slope(m), inorm=m._inorm)[0]

        h = x.float().contiguous()
        for i, ((name, _, _), attr) in enumerate(zip(_ENCODER, _SKIPS)):
            if i:
                h = ops.maxpool_forward(ops.to_act(h, dt), dt)
            st = getattr(self, name)
            h = lu(st.ops[1], lu(st.ops[0], h))
            setattr(self, attr, h)
        feats, masks = [], []
        for (name, _, _), factor in zip(_DECODER, _UPSAMPLE):
            up = getattr(self, name)
            h = lu(up.ops[1], lu(up.ops[0], ops.convt_forward(ops.to_act(h, dt), up.up_conv.weight, up.up_conv.bias, up._packed_up, dt)))
            ph = up.predictor_head
            pro = ops.bn1d_eval(ops.gap_forward(h, dt), up.bn.weight, up.bn.bias, up.bn.running_mean, up.bn.running_var, relu=False)
            hid = ops.bn1d_eval(ops.linear_forward(pro, ph[0].weight, ph[0].bias), ph[1].weight, ph[1].bias, ph[1].running_mean, ph[1].running_var, relu=True)
            feats.append([pro, ops.linear_forward(hid, ph[3].weight, ph[3].bias)])
            if not local:
                mask = lu(up.deep_supervision_head, h)
                masks.append(mask if factor == 1 else ops.upsample_forward(mask, factor))
        return ops.conv1x1_to1_forward(ops.to_act(h, dt), self.out_tr.final_conv.weight, self.out_tr.final_conv.bias, dt), feats, masks

    def _train_stages(self, x, local, pass_idx, features_only=False, lazy_skips=False):
        """The training-mode forward.  (Rounds 3-4 had this as a generator so that several passes could be advanced stage by stage in rotation,
        each on its own stream, with one matrix kernel at a time -- measured slower, 33.5 -> 34.4 / 37.1 ms, DESIGN section 5 -- removed.)
        features_only: the caller discards the reconstruction and the deep-supervision maps (the second view and the local views of a
        training step, train_3d.py:117,123 -- SURVEY Q3): `out_tr` (no state) and the trilinear upsampling are skipped and None / [] are
        returned in their place; everything that has STATE -- the deep-supervision heads' BatchNorm running statistics -- still runs."""
        mods = self._stage_modules()

        def mine():          # the stage Functions read the pass number off their module at forward time
            for m in mods:
                m._pass_idx = pass_idx

        h, pooled = x, None
        for i, ((name, _, _), attr) in enumerate(zip(_ENCODER, _SKIPS)):
            stage = getattr(self, name)
            mine()
            h = h if i == 0 else (pooled if pooled is not None else self.maxpool(h))
            last = stage.ops[1]
            if config.FOLD_POOL_GRAD and i + 1 < len(_ENCODER) and not last._gn_groups:
                # stage output and `self.maxpool` of it (:115-117) as one node: the pool's backward folds into the BatchNorm backward
                a = stage.ops[0](h)
                h, pooled = last.forward_pooled(a, pool_only=lazy_skips)     # lazy_skips: h is a _LazySkip (the stage output is not stored)
            else:
                h, pooled = stage(h), None
            setattr(self, attr, h)          # the reference keeps these alive as attributes; the skips are never consumed (D6)
        middle_features, middle_masks = [], []
        for (name, _, _), factor in zip(_DECODER, _UPSAMPLE):
            mine()
            h, pro, pre, mask = getattr(self, name)(h)
            middle_features.append([pro, pre])
            if not local and not features_only:
                middle_masks.append(mask if factor == 1 else Fn.TrilinearFn.apply(mask, factor))
        if features_only:
            return None, middle_features, middle_masks
        mine()
        out = self.out_tr(h)
        return out, middle_features, middle_masks

    def forward(self, x, local=False, *, features_only=False, lazy_skips=False):
        """-> (out [b,1,D,H,W], [[pro, pre] x 3 scales], [mask x 3] or [] when local).  `features_only` (engine extension, keyword only):
        (None, features, []) -- see _train_stages.  `lazy_skips` (engine extension, keyword only; train_3d.step_losses sets it): the encoder stages'
        unpooled outputs -- which the reference stashes as `self.skip_out64 / 128 / 256` (:114-117) and never reads -- are not written to HBM; the
        attributes still answer, computing the tensor from the stage's saved pre-normalisation values when read (detached)."""
        if not x.is_cuda:
            raise RuntimeError("PCRLv23d (pcrlv2_amd) runs on the GPU only: input is on %s and there is no CPU fallback" % x.device)
        if not self.training:
            return self._forward_eval(x, local)
        result = self._train_stages(x, local, ops.next_pass(), features_only, lazy_skips)     # pass 0 = first forward since the last optimizer step (its backward runs last)
        ops.end_of_forward_join()           # the stages' side branches (config.FWD_BRANCH_STREAM) are complete when the outputs are handed out
        return result


def _skip_attribute(name):
    """`model.skip_out64` etc.: plain attributes as in the reference (:114-117), except that a _LazySkip stored by the engine's training step is
    turned into its tensor when read."""
    def get(self):
        store = self.__dict__.get("_skip_store")
        if store is None or name not in store:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}' (set by forward())")
        v = store[name]
        if isinstance(v, _LazySkip):
            v = store[name] = v.materialize()
        return v

    def put(self, v):
        self.__dict__.setdefault("_skip_store", {})[name] = v
    return property(get, put)


for _n in _SKIPS[:3]:
    setattr(PCRLv23d, _n, _skip_attribute(_n))

