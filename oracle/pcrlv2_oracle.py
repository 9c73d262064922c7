"""CPU oracle for the PCRLv2 3D pre-training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional (no nn.Module) restatement, in plain PyTorch-CPU ops, of the
reference's algorithm for the path named by BASELINE.json:north_star.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
path (`pcrlv2_amd`) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  `oracle/make_golden.py` imports the real reference
(`/root/reference/models/pcrlv2_model_3d.py`, `train_3d.cos_loss`,
`utils.adjust_learning_rate`) in the authoring container, runs it in float64 with oneDNN
disabled on the closed-form inputs of `fill_*` below, checks this restatement against it
(forward, losses, every gradient, 2 SGD steps) and writes `tests/golden/*.npz`.
`tests/test_oracle_golden.py` re-checks the restatement against those vectors everywhere.

Every function cites the reference lines it follows (paths relative to the reference repo).
Layout: logical NCDHW like the reference; dtype follows the inputs (float64 or float32).
"""
from __future__ import annotations

import math
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# Model structure (models/pcrlv2_model_3d.py:37-45, 95-110): channel plan per LUConv.
# ----------------------------------------------------------------------------------------
ENCODER = [  # (prefix, Cin, Cout)   DownTransition(in, depth): in -> 32*2^d -> 64*2^d
    ("down_tr64.ops.0", None, 32), ("down_tr64.ops.1", 32, 64),
    ("down_tr128.ops.0", 64, 64), ("down_tr128.ops.1", 64, 128),
    ("down_tr256.ops.0", 128, 128), ("down_tr256.ops.1", 128, 256),
    ("down_tr512.ops.0", 256, 256), ("down_tr512.ops.1", 256, 512),
]
DECODER = [  # (name, C_up (= in = out of the transposed conv), C)   double_chnnel=True
    ("up_tr256", 512, 256), ("up_tr128", 256, 128), ("up_tr64", 128, 64),
]
BN_EPS = 1e-5       # nn.BatchNorm3d / BatchNorm1d default, pcrlv2_model_3d.py:12,55,57
BN_MOMENTUM = 0.1   # pcrlv2_model_3d.py:12


def state_layout(n_class: int = 1, in_channels: int = 1, act: str = "relu", norm: str = "bn") -> "OrderedDict[str, tuple]":
    """Names and shapes of the 169 state_dict entries, in registration order.

    Follows the module construction order of pcrlv2_model_3d.py:6-34 (LUConv: conv1, bn1, activation),
    :48-60 (UpTransition: up_conv, ops, bn, predictor_head, deep_supervision_head),
    :75-79 (OutputTransition) and :98-110 (PCRLv23d).
    Constructor variants (never instantiated by train_3d.py:45, accepted by the constructor): norm='in' -- InstanceNorm3d(affine=True)
    keeps weight and bias only (:16, no running statistics by default); act='prelu' -- nn.PReLU(out_chan) adds `activation.weight` after
    bn1 in every LUConv except the sigmoid heads (:22-23).
    """
    lay: "OrderedDict[str, tuple]" = OrderedDict()

    def luconv(p, ci, co, head=False):
        lay[p + ".conv1.weight"] = (co, ci, 3, 3, 3)
        lay[p + ".conv1.bias"] = (co,)
        if norm == "in":
            lay[p + ".bn1.weight"] = (co,)
            lay[p + ".bn1.bias"] = (co,)
        else:
            bn(p + ".bn1", co)
        if act == "prelu" and not head:
            lay[p + ".activation.weight"] = (co,)

    def bn(p, c):
        lay[p + ".weight"] = (c,)
        lay[p + ".bias"] = (c,)
        lay[p + ".running_mean"] = (c,)
        lay[p + ".running_var"] = (c,)
        lay[p + ".num_batches_tracked"] = ()

    for p, ci, co in ENCODER:
        luconv(p, in_channels if ci is None else ci, co)
    for name, cu, c in DECODER:
        lay[name + ".up_conv.weight"] = (cu, cu, 2, 2, 2)
        lay[name + ".up_conv.bias"] = (cu,)
        luconv(name + ".ops.0", cu, c)
        luconv(name + ".ops.1", c, c)
        bn(name + ".bn", c)
        lay[name + ".predictor_head.0.weight"] = (2 * c, c)
        lay[name + ".predictor_head.0.bias"] = (2 * c,)
        bn(name + ".predictor_head.1", 2 * c)
        lay[name + ".predictor_head.3.weight"] = (c, 2 * c)
        lay[name + ".predictor_head.3.bias"] = (c,)
        luconv(name + ".deep_supervision_head", c, 1, head=True)
    lay["out_tr.final_conv.weight"] = (n_class, 64, 1, 1, 1)
    lay["out_tr.final_conv.bias"] = (n_class,)
    return lay


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


# ----------------------------------------------------------------------------------------
# Closed-form deterministic fills (exact integer hash -> portable across machines).
# ----------------------------------------------------------------------------------------
def _hash_uniform(n: int, seed: int) -> np.ndarray:
    """n values in [-1, 1), from a splitmix64-style integer hash of (index, seed)."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def _name_seed(name: str) -> int:
    s = 1469598103934665603
    for ch in name.encode():
        s = ((s ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return s >> 8


def fill_state(dtype=torch.float64, n_class: int = 1, in_channels: int = 1, act: str = "relu", norm: str = "bn") -> "OrderedDict[str, torch.Tensor]":
    """Deterministic stand-in for PyTorch's default init (kaiming-uniform(a=sqrt 5) bounds:
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv/linear weight and bias; BN weight near 1,
    bias near 0 but not exactly so that their gradients are exercised)."""
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in state_layout(n_class, in_channels, act, norm).items():
        n = int(np.prod(shape)) if shape else 1
        u = _hash_uniform(n, _name_seed(name))
        if name.endswith("num_batches_tracked"):
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        if name.endswith("running_mean"):
            v = np.zeros(n)
        elif name.endswith("running_var"):
            v = np.ones(n)
        elif name.endswith("activation.weight"):
            v = 0.25 + 0.1 * u          # nn.PReLU initialises its slopes to 0.25
        elif ".bn" in name or "predictor_head.1" in name:
            v = (1.0 + 0.1 * u) if name.endswith("weight") else 0.1 * u
        else:
            if name.endswith("weight"):
                fan_in = int(np.prod(shape[1:]))
                if "up_conv" in name:  # ConvTranspose3d: fan_in computed from dim 1 * k^3 as well
                    fan_in = shape[1] * 8
            else:
                wshape = state_layout(n_class, in_channels, act, norm)[name[:-4] + "weight"]
                fan_in = int(np.prod(wshape[1:]))
                if "up_conv" in name:
                    fan_in = wshape[1] * 8
            v = u / math.sqrt(fan_in)
        st[name] = torch.from_numpy(v.reshape(shape)).to(dtype)
    return st


def fill_batch(b: int, dhw=(32, 32, 16), local=16, dtype=torch.float64, seed: int = 7):
    """Synthetic batch with the input contract of datasets/lunaDataset.py:79-81:
    (input1, input2, gt, gt2, [6 local views]).  Views are correlated (x2 = x1 + 0.25*noise,
    locals = crops of x1 + noise) so the cosine terms are well conditioned."""
    D, H, W = dhw
    n = b * D * H * W
    x1 = _hash_uniform(n, seed).reshape(b, 1, D, H, W) * 1.7
    x2 = x1 + 0.25 * _hash_uniform(n, seed + 1).reshape(b, 1, D, H, W)
    gt = 0.5 + 0.5 * _hash_uniform(n, seed + 2).reshape(b, 1, D, H, W)
    gt2 = 0.5 + 0.5 * _hash_uniform(n, seed + 3).reshape(b, 1, D, H, W)
    locs = []
    for i in range(6):
        d0, h0, w0 = (i * 5) % (D - local + 1), (i * 7) % (H - local + 1), (i * 3) % (W - local + 1)
        crop = x1[:, :, d0:d0 + local, h0:h0 + local, w0:w0 + local]
        noise = _hash_uniform(b * local ** 3, seed + 10 + i).reshape(b, 1, local, local, local)
        locs.append(torch.from_numpy(np.ascontiguousarray(crop + 0.25 * noise)).to(dtype))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return t(x1), t(x2), t(gt), t(gt2), locs


# ----------------------------------------------------------------------------------------
# bf16-EXACT operator cases for the tight pin of the MFMA kernels (oracle/make_golden.py --mfma-pin,
# tests/test_mfma_pin_gpu.py).  Every operand is exactly representable in bfloat16, so a bf16 MFMA kernel forms exact products and
# differs from a float64 evaluation only by float32 accumulation order: its float32 outputs (BatchNorm partial statistics, weight
# gradients) can be held at float32-class tolerance, its bf16 outputs at "equal to the correctly rounded reference, up to one ulp on
# the rare round-off ties".
# ----------------------------------------------------------------------------------------
MFMA_PIN_CONV = {      # name -> (N, (D, H, W), Ci, Co): all tile into 4 x 8 x 16 bricks (the last: along (D, W, H) -- the PERM instantiations)
    "lu_32_64": (2, (8, 16, 32), 32, 64),
    "lu_128_64": (1, (4, 8, 16), 128, 64),
    "lu_64_128": (1, (8, 16, 16), 64, 128),
    "lu_64_64_perm": (1, (8, 32, 8), 64, 64),
}
MFMA_PIN_UP = {        # name -> (N, coarse (D, H, W), C of UpTransition(C, C, 0): ConvTranspose3d(C -> C) into LUConv(C -> 64))
    "up_128": (1, (4, 8, 16), 128),
    "up_64_perm": (2, (4, 16, 8), 64),
}


def _bf16_exact(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).double()


def mfma_pin_conv_case(name: str):
    """-> dict of float64 tensors, all bf16-representable: x [N,Ci,D,H,W], w [Co,Ci,3,3,3], dy [N,Co,D,H,W]; plus float64 b, gamma, beta [Co]."""
    N, (D, H, W), Ci, Co = MFMA_PIN_CONV[name]
    sd = _name_seed(name) % 100000
    x = _bf16_exact(_hash_uniform(N * Ci * D * H * W, sd).reshape(N, Ci, D, H, W) * 1.5)
    w = _bf16_exact(_hash_uniform(Co * Ci * 27, sd + 1).reshape(Co, Ci, 3, 3, 3) * (1.2 / math.sqrt(27 * Ci)))
    dy = _bf16_exact(_hash_uniform(N * Co * D * H * W, sd + 2).reshape(N, Co, D, H, W))
    f = lambda n, k, lo, sc: torch.from_numpy(lo + sc * _hash_uniform(n, sd + k)).double()
    return dict(x=x, w=w, dy=dy, b=f(Co, 3, 0.0, 0.2), gamma=f(Co, 4, 1.0, 0.3), beta=f(Co, 5, 0.0, 0.3))


def mfma_pin_up_case(name: str):
    """Weights of ConvTranspose3d(C -> C, k2 s2) and Conv3d(C -> 64, 3x3x3) whose COMPOSED 8-tap phase weights are exactly
    representable in bf16: the transposed convolution is sparse (two non-zero intermediate channels per (input channel, position), values
    +-1/4, +-1/2), the 3x3x3 weights are small integers / 32 -- every composed entry is k / 128 with |k| <= 64.
    -> x [N,C,D,H,W] (bf16-exact), w_up [C,C,2,2,2], b_up [C], w0 [64,C,3,3,3], b0 [64], dy0 [N,64,2D,2H,2W] (bf16-exact)."""
    N, (D, H, W), C = MFMA_PIN_UP[name]
    Co = 64
    sd = _name_seed(name) % 100000
    x = _bf16_exact(_hash_uniform(N * C * D * H * W, sd).reshape(N, C, D, H, W) * 1.5)
    u = _hash_uniform(C * 8 * 4, sd + 1).reshape(C, 8, 4)
    w_up = np.zeros((C, C, 8))
    for ci in range(C):
        for s in range(8):
            c0 = int((u[ci, s, 0] * 0.5 + 0.5) * C) % C
            c1 = (c0 + 1 + int((u[ci, s, 1] * 0.5 + 0.5) * (C - 1))) % C
            w_up[ci, c0, s] = (0.25 if u[ci, s, 2] < 0 else 0.5) * (1 if u[ci, s, 3] < 0 else -1)
            w_up[ci, c1, s] = (0.5 if u[ci, s, 2] < 0.3 else 0.25) * (-1 if u[ci, s, 3] < 0.2 else 1)
    w_up = torch.from_numpy(w_up.reshape(C, C, 2, 2, 2)).double()
    w0 = torch.from_numpy(np.round(_hash_uniform(Co * C * 27, sd + 2).reshape(Co, C, 3, 3, 3) * 2.49) / 32.0).double()
    dy0 = _bf16_exact(_hash_uniform(N * Co * 8 * D * H * W, sd + 3).reshape(N, Co, 2 * D, 2 * H, 2 * W))
    f = lambda n, k, sc: torch.from_numpy(sc * _hash_uniform(n, sd + k)).double()
    return dict(x=x, w_up=w_up, b_up=f(C, 4, 0.3), w0=w0, b0=f(Co, 5, 0.2), dy0=dy0)


# ----------------------------------------------------------------------------------------
# Forward pass
# ----------------------------------------------------------------------------------------
_CFG = {"act": "relu", "norm": "bn"}   # constructor variant in force (forward(act=, norm=)); the default is what train_3d.py:45 instantiates
_EVAL = False   # set by forward(training=False): batch norms use the running statistics (nn.Module.eval() semantics)


def _bn_train(x, st, p, new_bufs):
    """Training-mode batch norm (any rank, channel dim 1): biased batch variance for the
    normalisation, unbiased for running_var, momentum 0.1 (nn.BatchNorm{1,3}d defaults as
    used at pcrlv2_model_3d.py:12,55,57).  In eval mode (forward(training=False)): running statistics, no update."""
    if _EVAL:
        shp = [1, -1] + [1] * (x.dim() - 2)
        rm, rv = st[p + ".running_mean"].to(x.dtype), st[p + ".running_var"].to(x.dtype)
        return (x - rm.view(shp)) / torch.sqrt(rv.view(shp) + BN_EPS) * st[p + ".weight"].view(shp) + st[p + ".bias"].view(shp)
    dims = [0] + list(range(2, x.dim()))
    m = x.numel() // x.shape[1]
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    shp = [1, -1] + [1] * (x.dim() - 2)
    y = (x - mean.view(shp)) / torch.sqrt(var.view(shp) + BN_EPS) * st[p + ".weight"].view(shp) + st[p + ".bias"].view(shp)
    if new_bufs is not None:
        with torch.no_grad():
            rm = new_bufs.get(p + ".running_mean", st[p + ".running_mean"])
            rv = new_bufs.get(p + ".running_var", st[p + ".running_var"])
            nb = new_bufs.get(p + ".num_batches_tracked", st[p + ".num_batches_tracked"])
            new_bufs[p + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach().to(rm.dtype)
            new_bufs[p + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * (var.detach() * (m / max(m - 1, 1))).to(rv.dtype)
            new_bufs[p + ".num_batches_tracked"] = nb + 1
    return y


def _luconv(x, st, p, new_bufs, act="relu"):
    """LUConv.forward, pcrlv2_model_3d.py:32-34: act(bn1(conv1(x))); conv 3x3x3 pad 1 with bias (:9)."""
    y = F.conv3d(x, st[p + ".conv1.weight"], st[p + ".conv1.bias"], padding=1)
    if _CFG["norm"] == "in":
        # norm='in' (pcrlv2_model_3d.py:15-16): InstanceNorm3d(affine=True), per-(sample, channel) statistics in train and eval mode alike
        y = F.instance_norm(y, weight=st[p + ".bn1.weight"], bias=st[p + ".bn1.bias"], eps=1e-5)
        return _activation(y, st, p, act)
    if (p + ".bn1.running_mean") not in st:
        # OPTIONAL non-reference mode of the engine (PCRLv23d(norm='gn', act='silu'), north_star's GroupNorm + SiLU): the state has
        # no running statistics for this layer -> GroupNorm(8) + SiLU.  Checked against torch's own F.group_norm / F.silu only.
        return F.silu(F.group_norm(y, 8, st[p + ".bn1.weight"], st[p + ".bn1.bias"], eps=1e-5))
    y = _bn_train(y, st, p + ".bn1", new_bufs)
    return _activation(y, st, p, act)


def _activation(y, st, p, act):
    """LUConv's activation, pcrlv2_model_3d.py:20-30: the sigmoid heads keep their sigmoid; every other LUConv takes the constructor's `act`."""
    if act == "sigmoid":
        return torch.sigmoid(y)
    a = _CFG["act"]
    if a == "relu":
        return torch.relu(y)
    if a == "elu":
        return F.elu(y)                                   # nn.ELU(): alpha = 1 (:25)
    if a == "prelu":
        return F.prelu(y, st[p + ".activation.weight"])   # nn.PReLU(out_chan): one slope per channel (:23)
    raise ValueError(a)


def _up_transition(x, st, name, new_bufs):
    """UpTransition.forward, pcrlv2_model_3d.py:62-72 (skip concat is commented out at :65)."""
    b = x.shape[0]
    up = F.conv_transpose3d(x, st[name + ".up_conv.weight"], st[name + ".up_conv.bias"], stride=2)   # :64
    x = _luconv(_luconv(up, st, name + ".ops.0", new_bufs), st, name + ".ops.1", new_bufs)            # :66
    x_pro = x.mean(dim=(2, 3, 4)).view(b, -1)                                                          # :67-68
    x_pro = _bn_train(x_pro, st, name + ".bn", new_bufs)                                              # :69
    h = F.linear(x_pro, st[name + ".predictor_head.0.weight"], st[name + ".predictor_head.0.bias"])    # :56
    h = torch.relu(_bn_train(h, st, name + ".predictor_head.1", new_bufs))                            # :57-58
    x_pre = F.linear(h, st[name + ".predictor_head.3.weight"], st[name + ".predictor_head.3.bias"])    # :59
    x_mask = _luconv(x, st, name + ".deep_supervision_head", new_bufs, act="sigmoid")                 # :71
    return x, x_pro, x_pre, x_mask


def forward(st, x, local: bool = False, new_bufs=None, training: bool = True, act: str = "relu", norm: str = "bn"):
    """PCRLv23d.forward, pcrlv2_model_3d.py:112-133.  `st` maps state_dict names to tensors
    (parameters may require grad).  Returns (out, [[pro,pre]x3], [mask x3] or []).
    `new_bufs` (dict) receives the updated BN running statistics, in call order.
    training=False: the module in .eval() mode (what a consumer of the checkpoint runs for validation, README.md:48-55)."""
    global _EVAL
    if (act, norm) != (_CFG["act"], _CFG["norm"]):
        keep = dict(_CFG)
        _CFG.update(act=act, norm=norm)
        try:
            return forward(st, x, local, new_bufs, training, act, norm)
        finally:
            _CFG.update(keep)
    if not training:
        _EVAL = True
        try:
            return forward(st, x, local, None, True, act, norm)
        finally:
            _EVAL = False
    h = x
    for i, (p, _, _) in enumerate(ENCODER):
        if i in (2, 4, 6):
            h = F.max_pool3d(h, 2)                                    # :115-117
        h = _luconv(h, st, p, new_bufs)
    feats, masks_raw = [], []
    for name, _, _ in DECODER:                                        # :120-123
        h, pro, pre, mk = _up_transition(h, st, name, new_bufs)
        feats.append([pro, pre])
        masks_raw.append(mk)
    masks = []
    if not local:                                                     # :124-127
        masks.append(F.interpolate(masks_raw[0], scale_factor=4, mode="trilinear"))
        masks.append(F.interpolate(masks_raw[1], scale_factor=2, mode="trilinear"))
        masks.append(masks_raw[2])
    out = torch.sigmoid(F.conv3d(h, st["out_tr.final_conv.weight"], st["out_tr.final_conv.bias"]))  # :78-82,132
    return out, feats, masks


def variant_input(b: int, dhw, in_channels: int, dtype=torch.float64, seed: int = 41):
    """[b, in_channels, D, H, W] closed-form input for the constructor-variant fixtures (tests/golden/v_*.npz)."""
    D, H, W = dhw
    x = _hash_uniform(b * in_channels * D * H * W, seed).reshape(b, in_channels, D, H, W) * 1.7
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def variant_loss(out, feats, masks, seed: int = 97):
    """A scalar that reaches every output of PCRLv23d.forward with fixed closed-form weights: sum over outputs of mean(output * R).
    Used by the constructor-variant fixtures (make_golden.make_variant) and their GPU tests to compare gradients of every parameter."""
    def term(t, k):
        r = torch.from_numpy(_hash_uniform(t.numel(), seed + k).reshape(tuple(t.shape))).to(device=t.device, dtype=t.dtype)
        return (t * r).mean()
    L = term(out, 0)
    for i, (pro, pre) in enumerate(feats):
        L = L + term(pro, 1 + 3 * i) + term(pre, 2 + 3 * i)
    for i, m in enumerate(masks):
        L = L + term(m, 3 + 3 * i)
    return L


# ----------------------------------------------------------------------------------------
# Losses and the training step
# ----------------------------------------------------------------------------------------
def cosine_similarity(x, y, eps: float = 1e-8):
    """nn.CosineSimilarity(dim=1, eps=1e-8) as instantiated at train_3d.py:57:
    sum(x*y) / (max(||x||, eps) * max(||y||, eps)) per row."""
    nx = x.norm(dim=1).clamp_min(eps)
    ny = y.norm(dim=1).clamp_min(eps)
    return (x * y).sum(dim=1) / (nx * ny)


def cos_loss(output1, output2, rng: random.Random):
    """train_3d.py:86-92.  One scale drawn with randint(0, len-1); symmetric negative cosine
    with stop-gradient on the projection (`pro`, index 0); predictor (`pre`) is index 1."""
    index = rng.randint(0, len(output1) - 1)
    s1, s2 = output1[index], output2[index]
    loss = -(cosine_similarity(s1[1], s2[0].detach()).mean() + cosine_similarity(s2[1], s1[0].detach()).mean()) * 0.5
    return loss, index


def step_losses(st, batch, epoch: int, rng: random.Random, new_bufs=None):
    """Forward half of one iteration of train_3d.py:109-138.  Returns a dict with
    loss (total), loss1 (MSE), loss2 (global cosine), loss4 (deep supervision), local_loss,
    index2 and the three forward outputs of view 1."""
    input1, input2, gt, _gt2, local_views = batch                      # :109 (gt2 never used)
    bsz = input1.size(0)
    mask1, dec1, mid1 = forward(st, input1, new_bufs=new_bufs)        # :116
    _mask2, dec2, _ = forward(st, input2, new_bufs=new_bufs)          # :117
    loss2, index2 = cos_loss(dec1, dec2, rng)                         # :119
    local_input = torch.cat(local_views, dim=0)                       # :121
    _, lout, _ = forward(st, local_input, local=True, new_bufs=new_bufs)  # :123
    lout = [torch.stack(t) for t in lout]                             # :125
    local_loss = 0.0
    for i in range(len(local_views)):                                 # :127-133
        tmp = [t[:, bsz * i: bsz * (i + 1)] for t in lout]
        l1, _ = cos_loss(dec1, tmp, rng)
        l2, _ = cos_loss(dec2, tmp, rng)
        local_loss = local_loss + l1 + l2
    local_loss = local_loss / (2 * len(local_views))                  # :134
    loss1 = F.mse_loss(mask1, gt)                                     # :135
    beta = 0.5 * (1.0 + math.cos(math.pi * epoch / 240))              # :136 (240 hard-coded)
    loss4 = beta * F.mse_loss(mid1[index2], gt)                       # :137
    loss = loss1 + loss2 + loss4 + local_loss                         # :138
    return dict(loss=loss, loss1=loss1, loss2=loss2, loss4=loss4, local_loss=local_loss,
                index2=index2, mask1=mask1, dec1=dec1, mid1=mid1)


def lr_at(epoch: int, base_lr: float, epochs: int) -> float:
    """utils.py:101-114: per-epoch cosine schedule."""
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * epoch / epochs))


def sgd_step(st, grads, mom, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD as configured at train_3d.py:48-51 (dampening 0, no nesterov, weight
    decay on every parameter): g += wd*p; buf = g (first step) | m*buf + g; p -= lr*buf."""
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            g = g + weight_decay * st[k]
            if k not in mom:
                mom[k] = g.clone()
            else:
                mom[k] = momentum * mom[k] + g
            st[k] = st[k] - lr * mom[k]
    return st, mom


def train_steps(st, batches, epoch=0, base_lr=1e-3, epochs=240, seed=0, momentum=0.9, weight_decay=1e-4):
    """k iterations of train_3d.py:109-151 from state `st` (one batch per iteration).
    Returns (final state, momentum buffers, list of per-step dicts of python floats,
    gradients of the first step)."""
    rng = random.Random(seed)
    st = OrderedDict((k, v.clone()) for k, v in st.items())
    mom, log, first_grads = {}, [], None
    lr = lr_at(epoch, base_lr, epochs)                                # train_3d.py:62
    for batch in batches:
        pnames = [k for k in st if not is_buffer(k)]
        for k in pnames:
            st[k] = st[k].detach().requires_grad_(True)
        new_bufs = {}
        r = step_losses(st, batch, epoch, rng, new_bufs)
        gl = torch.autograd.grad(r["loss"], [st[k] for k in pnames], allow_unused=True)  # :146
        grads = dict(zip(pnames, gl))
        if first_grads is None:
            first_grads = {k: (None if g is None else g.detach().clone()) for k, g in grads.items()}
        for k in pnames:
            st[k] = st[k].detach()
        st, mom = sgd_step(st, grads, mom, lr, momentum, weight_decay)  # :151
        st.update(new_bufs)
        log.append({k: float(r[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")} | {"index2": r["index2"]})
    return st, mom, log, first_grads


def step_losses_data_parallel_chunked(st, per_rank, epoch: int, rng: random.Random):
    """One iteration of train_3d.py:109-138 under nn.DataParallel with the LITERAL scatter of the local views (train_3d.py:121-123): the
    six local views of the GLOBAL batch are concatenated view-major into one [6B] tensor and `model(local_input, local=True)` scatters THAT
    along dim 0 -- replica r runs rows [r * 6B / W, (r + 1) * 6B / W), i.e. with two replicas replica 0 sees local views 0-2 of ALL samples
    (its BatchNorm statistics of the local pass are over those rows), the outputs are gathered back into [6B] order and the losses are
    taken over the gathered batch.  per_rank[r] = replica r's share of the global batch (global sample order = replica 0's samples, then
    replica 1's, ...).  -> (list of per-replica result dicts with the loss restricted to the replica's own samples -- their mean is the
    reference's loss over the gathered batch --, replica 0's new buffers)."""
    W = len(per_rank)
    b = per_rank[0][0].size(0)
    B = W * b
    nl = len(per_rank[0][4])
    draws = rng.getstate()
    bufs = [dict() for _ in range(W)]
    fw = []
    for r, batch in enumerate(per_rank):            # every replica: view 1, then view 2 (the order replica 0's running statistics see)
        mask1, dec1, mid1 = forward(st, batch[0], new_bufs=bufs[r])
        fw.append([mask1, dec1, mid1, None])
    for r, batch in enumerate(per_rank):
        fw[r][3] = forward(st, batch[1], new_bufs=bufs[r])[1]
    local_global = torch.cat([torch.cat([per_rank[r][4][v] for r in range(W)], dim=0) for v in range(nl)], dim=0)     # [6B]: view-major, samples in global order
    rows = nl * B // W
    louts = []
    for r in range(W):
        _, lo, _ = forward(st, local_global[r * rows:(r + 1) * rows], local=True, new_bufs=bufs[r])
        louts.append(lo)
    lout = [torch.stack([torch.cat([louts[r][k][j] for r in range(W)], dim=0) for j in range(2)]) for k in range(len(louts[0]))]   # gathered: [2, 6B, C] per scale
    out = []
    for r, batch in enumerate(per_rank):
        rng.setstate(draws)                          # ONE draw per cos_loss call for the whole gathered batch
        mask1, dec1, mid1, dec2 = fw[r]
        loss2, index2 = cos_loss(dec1, dec2, rng)
        local_loss = 0.0
        for i in range(nl):
            tmp = [t[:, B * i + r * b: B * i + (r + 1) * b] for t in lout]      # this replica's samples of local view i
            l1, _ = cos_loss(dec1, tmp, rng)
            l2, _ = cos_loss(dec2, tmp, rng)
            local_loss = local_loss + l1 + l2
        local_loss = local_loss / (2 * nl)
        loss1 = F.mse_loss(mask1, batch[2])
        beta = 0.5 * (1.0 + math.cos(math.pi * epoch / 240))
        loss4 = beta * F.mse_loss(mid1[index2], batch[2])
        out.append(dict(loss=loss1 + loss2 + loss4 + local_loss, loss1=loss1, loss2=loss2, loss4=loss4, local_loss=local_loss, index2=index2))
    return out, bufs[0]


def train_steps_data_parallel(st, rank_batches, epoch=0, base_lr=1e-3, epochs=240, seed=0, momentum=0.9, weight_decay=1e-4, partition="sample"):
    """k iterations under the reference's `nn.DataParallel` (train_3d.py:54), restated for `world` replicas.

    rank_batches[s][r] = the batch of replica r in iteration s (DataParallel scatters the global batch along dim 0, one chunk per
    replica).  Semantics followed (torch.nn.parallel.DataParallel as train_3d.py:54 uses it):
      * every replica runs the whole module on its chunk with the SAME parameters -- BatchNorm3d / BatchNorm1d statistics are
        per replica (pcrlv2_model_3d.py:12,55,57 in train mode);
      * replica 0 is the module itself: only ITS running statistics persist (`replicate` hands device 0 the original buffers);
      * the outputs are gathered and the losses (train_3d.py:119-138) are means over the gathered global batch -- with equal chunks
        the mean of the replicas' losses; the scales are drawn ONCE per cos_loss call for the whole batch (every replica sees the
        same draw);
      * one backward, gradients of all replicas summed into the one parameter set, one SGD step (train_3d.py:143-151).
    partition="sample" (the engine's default, one deliberate difference stated in DESIGN.md): the local views are partitioned BY SAMPLE (each
    replica holds all six local views of its own crops, as one-process-per-GPU loaders deliver them); partition="chunk": by chunks of the
    concatenated [6B] tensor, as nn.DataParallel literally does (PCRL_DP_LOCAL_PARTITION=chunk in the engine).
    Returns (final state with replica 0's buffers, momentum, per-step list of per-replica loss dicts, gradients of the first step)."""
    rng = random.Random(seed)
    st = OrderedDict((k, v.clone()) for k, v in st.items())
    mom, log, first_grads = {}, [], None
    lr = lr_at(epoch, base_lr, epochs)
    for per_rank in rank_batches:
        pnames = [k for k in st if not is_buffer(k)]
        for k in pnames:
            st[k] = st[k].detach().requires_grad_(True)
        draws = rng.getstate()
        total, bufs0, entry = 0.0, None, []
        if partition == "chunk":        # the reference's literal scatter of the concatenated local views (step_losses_data_parallel_chunked)
            results, bufs0 = step_losses_data_parallel_chunked(st, per_rank, epoch, rng)
            for res in results:
                total = total + res["loss"] / len(per_rank)
                entry.append({k: float(res[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")} | {"index2": res["index2"]})
        for r, batch in enumerate(per_rank if partition != "chunk" else []):
            rng.setstate(draws)                     # one draw per cos_loss call for the whole gathered batch
            new_bufs = {}
            res = step_losses(st, batch, epoch, rng, new_bufs)
            total = total + res["loss"] / len(per_rank)
            entry.append({k: float(res[k].detach()) for k in ("loss", "loss1", "loss2", "loss4", "local_loss")} | {"index2": res["index2"]})
            if r == 0:
                bufs0 = new_bufs
        gl = torch.autograd.grad(total, [st[k] for k in pnames], allow_unused=True)
        grads = dict(zip(pnames, gl))
        if first_grads is None:
            first_grads = {k: (None if g is None else g.detach().clone()) for k, g in grads.items()}
        for k in pnames:
            st[k] = st[k].detach()
        st, mom = sgd_step(st, grads, mom, lr, momentum, weight_decay)
        st.update(bufs0)
        log.append(entry)
    return st, mom, log, first_grads
