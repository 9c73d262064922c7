"""What a SET of launches costs inside the three-stream C2 step: run them twice (they are idempotent) and take the step-time difference.

One-stream kernel sums overstate what removing a launch set would buy: in the three-stream step HBM- and latency-bound kernels run under other streams'
matrix kernels.  This probe measures the set's cost where it counts, without building the fused form first: blocks of K steps alternate between the
plain step and the step with the set doubled, same process, same box, same draws.

    python tools/double_ablation.py [--steps 12] [--rounds 3]
Sets: finalize (pcrl_bn_finalize + pcrl_bn_bwd_finalize: 94 launches), bn_reduce (every pcrl_bn_act_bwd_reduce*: the HBM-bound first pass),
small_conv (forward / data gradient convolutions of the 4^3 / 2^3 local-view levels), wgrad_reduce (the fixed-order second passes of the
weight gradients, through pcrl_debug_set_reduce_repeat)."""
import argparse
import os
import random
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import synthetic_batch  # noqa: E402
from pcrlv2_amd import _lib  # noqa: E402
from pcrlv2_amd.models import PCRLv23d  # noqa: E402
from pcrlv2_amd.optim import FusedSGD  # noqa: E402
from pcrlv2_amd.train_3d import CosineSimilarityMean, MSELoss, train_step  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--d", type=int, default=3, choices=[2, 3], help="2: the C5 per-GPU 2D step (512 x 512, b = 64) with its own launch sets")
a = ap.parse_args()
dev = torch.device("cuda")
L = _lib.lib()
L.cdll.pcrl_debug_set_reduce_repeat.argtypes = [__import__("ctypes").c_int]
L.cdll.pcrl_debug_set_reduce_repeat.restype = None


def small(name, args):   # convolution launches on volumes of <= 64 voxels
    if name in ("pcrl_conv3d_k3_fwd_ws",):
        return args[8] * args[9] * args[10] <= 64
    if name == "pcrl_upconv_fwd":
        return args[7] * args[8] * args[9] <= 64
    if name == "pcrl_upconv_dgrad_ws":
        return args[7] * args[8] * args[9] <= 64
    return False


def big_conv(name, args):   # plain forward / data gradient convolutions on volumes of > 64 voxels (the brick kernels)
    if name == "pcrl_conv3d_k3_fwd_ws":
        return args[8] * args[9] * args[10] > 64
    return name == "pcrl_conv3d_k3_dgrad_bnred"


def level_8x8x4(name, args):   # plain forward / data gradient convolutions of the GLOBAL views' 8 x 8 x 4 level (down_tr512, up_tr256.ops): VERDICT r5 item 6
    if name == "pcrl_conv3d_k3_fwd_ws":
        return args[8] * args[9] * args[10] == 256 and args[7] <= 64
    if name == "pcrl_conv3d_k3_dgrad_bnred":
        return args[10] * args[11] * args[12] == 256 and args[9] <= 64
    return False


SETS = {
    "level_8x8x4": level_8x8x4,
    "level_8x8x4_all": lambda n, ar: level_8x8x4(n, ar) or (n in ("pcrl_upconv_fwd", "pcrl_upconv_dgrad_ws") and ar[7] * ar[8] * ar[9] == 256),
    "finalize": lambda n, ar: n in ("pcrl_bn_finalize", "pcrl_bn_bwd_finalize"),
    "bn_reduce": lambda n, ar: n.startswith("pcrl_bn_act_bwd_reduce"),
    "bn_bwd_apply": lambda n, ar: n.startswith("pcrl_bn_act_bwd_apply"),
    "bn_apply": lambda n, ar: n in ("pcrl_bn_act_apply", "pcrl_bn_act_apply_pool", "pcrl_bn_act_apply_gap"),
    "small_conv": small,
    "big_conv": big_conv,
    "upconv_fwd_dgrad": lambda n, ar: n in ("pcrl_upconv_fwd", "pcrl_upconv_dgrad_ws") and not small(n, ar),
    "wgrad": lambda n, ar: n == "pcrl_conv3d_k3_wgrad",
    "to1_heads": lambda n, ar: n in ("pcrl_conv3d_to1_fwd", "pcrl_conv3d_to1_dgrad", "pcrl_conv3d_to1_wgrad"),
    "first_layer": lambda n, ar: n in ("pcrl_conv3d_k3_c1_fwd", "pcrl_conv3d_k3_c1_wgrad"),
    "upc_compose": lambda n, ar: n == "pcrl_upconv_compose",
    "upc_wgrad_accum": lambda n, ar: n == "pcrl_upconv_wgrad_accum",      # (accumulates: the values double, the timing is what is measured)
    "upc_wgrad_finish": lambda n, ar: n == "pcrl_upconv_wgrad_finish",
    "wgrad_reduce": None,
    "no_bnred": None,
}
if os.environ.get("ABL_SETS"):
    SETS = {k: v for k, v in SETS.items() if k in os.environ["ABL_SETS"].split(",")}
active = [None]
orig_call = _lib._Lib.call


def call(self, name, *args):
    r = orig_call(self, name, *args)
    f = active[0]
    if f is not None and f(name, args):
        orig_call(self, name, *args)
    return r


_lib._Lib.call = call
torch.manual_seed(0)
if a.d == 3:
    model = PCRLv23d().to(dev).train().set_compute_dtype(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = synthetic_batch(32, (64, 64, 32), 16, dev, 1234)
    crit, cosine = MSELoss(), CosineSimilarityMean()
else:
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.models import PCRLv2
    model = PCRLv2().cuda().set_compute_dtype("bf16")
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    x1 = torch.randn(64, 3, 512, 512, generator=g)
    batch = (x1.to(dev), (x1 + 0.1 * torch.randn(64, 3, 512, 512, generator=g)).to(dev), torch.rand(64, 3, 512, 512, generator=g).to(dev), None,
             [torch.randn(64, 3, 96, 96, generator=g).to(dev) for _ in range(6)])
    crit, cosine = train_2d.MSELoss2d(), CosineSimilarityMean()
    _step2d = train_2d.train_step

    def train_step(model, opt, batch, epoch, crit, cosine, guard=False):   # noqa: F811  (the 2D step behind the 3D step's name)
        return _step2d(model, opt, batch, epoch, crit, cosine)

    SETS = {
        "finalize": lambda n, ar: n in ("pcrl_bn_finalize", "pcrl_bn_bwd_finalize"),
        "bn_reduce": lambda n, ar: n.startswith("pcrl_bn_act_bwd_reduce"),
        "bn_bwd_apply": lambda n, ar: n.startswith("pcrl_bn_act_bwd_apply"),
        "bn_apply": lambda n, ar: n in ("pcrl_bn_act_apply", "pcrl_bn_act_apply_gap", "pcrl_bn_add_relu_fwd", "pcrl_bn_relu_maxpool2d_3s2_fwd"),
        "conv_fwd": lambda n, ar: n == "pcrl_conv2d_fwd",
        "conv_dgrad": lambda n, ar: n in ("pcrl_conv2d_dgrad", "pcrl_conv2d_dgrad_bnred", "pcrl_conv2d_dgrad_up", "pcrl_conv2d_dgrad_s2"),
        "conv_wgrad": lambda n, ar: n == "pcrl_conv2d_wgrad",
        "weight_packs": lambda n, ar: n in ("pcrl_conv2d_pack", "pcrl_conv2d_pack_s2", "pcrl_stem7_pack"),
        "stem": lambda n, ar: n in ("pcrl_stem7_fwd", "pcrl_stem7_wgrad"),
        "masks_pools": lambda n, ar: n in ("pcrl_relu_mask_sum_bwd", "pcrl_relu_mask_bwd", "pcrl_maxpool2d_3s2_bwd_sum", "pcrl_upsample2d_nearest2_bwd"),
        "no_bnred": None,
    }
    if os.environ.get("ABL_SETS"):
        SETS = {k: v for k, v in SETS.items() if k in os.environ["ABL_SETS"].split(",")}
random.seed(0)
for _ in range(8 if a.d == 3 else 4):
    train_step(model, opt, batch, 0, crit, cosine, guard=False)
st0 = random.getstate()


def block(which):
    from pcrlv2_amd import config
    if which == "no_bnred":           # not a doubling: the separate first BatchNorm-backward pass instead of the data gradient's fused one (config.DGRAD_BNRED)
        config.DGRAD_BNRED = False
    elif which == "wgrad_reduce":
        L.cdll.pcrl_debug_set_reduce_repeat(2)
    elif which is not None:
        active[0] = SETS[which]
    random.setstate(st0)
    for _ in range(2):
        train_step(model, opt, batch, 0, crit, cosine, guard=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        train_step(model, opt, batch, 0, crit, cosine, guard=False)
    e1.record()
    torch.cuda.synchronize()
    active[0] = None
    config.DGRAD_BNRED = True
    L.cdll.pcrl_debug_set_reduce_repeat(1)
    return e0.elapsed_time(e1) / a.steps


res = {k: [] for k in [None, *SETS]}
for _ in range(a.rounds):
    for k in res:
        res[k].append(block(k))
base = sum(res[None]) / a.rounds
print(f"plain step            : {['%.3f' % v for v in res[None]]}  mean {base:.3f} ms")
for k in SETS:
    m = sum(res[k]) / a.rounds
    print(f"{k:12s} doubled  : {['%.3f' % v for v in res[k]]}  mean {m:.3f} ms  -> the set costs {m - base:+.3f} ms inside the step")
