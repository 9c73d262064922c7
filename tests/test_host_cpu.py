"""CPU: host logic, the C-ABI surface and the drop-in API (no GPU compute)."""
import ctypes
import math
import random
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from pcrlv2_amd import _lib
    if not os.path.exists(_lib.LIBPATH):
        import __graft_entry__ as g
        g.build()
    protos = _lib.parse_header()
    assert len(protos) >= 44
    cdll = ctypes.CDLL(_lib.LIBPATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/pcrl_hip.h but not exported"
    L = _lib.lib()
    assert "gfx950" in L.version()
    # workspace-size helpers are pure host functions: callable without a GPU
    assert L.call("pcrl_conv3d_k3_wgrad_ws_bytes", 2, 16, 16, 16, 64, 64) > 0
    assert L.call("pcrl_bn_bwd_partial_rows", 1 << 22) == 4096      # 1024-row tiles on the big volumes
    assert L.call("pcrl_bn_bwd_partial_rows", 5000) == 157          # 32-row tiles: small volumes still spread over the chip
    assert L.call("pcrl_conv3d_k3_fwd_ws_bytes", 32, 8, 8, 4, 256, 256, 1) == 0   # 8x8x4 bottleneck level in bf16: brick kernel (axis permutation)
    assert L.call("pcrl_conv3d_k3_fwd_ws_bytes", 32, 8, 8, 4, 256, 256, 0) > 0    # float32 parity mode: gather kernel with split-K
    assert L.call("pcrl_conv3d_k3_fwd_ws_bytes", 192, 4, 4, 4, 256, 256, 1) > 0   # 4^3 level of the local views: split-K
    assert L.call("pcrl_conv3d_k3_stats_rows", 32, 64, 64, 32, 64, 64, 1) == 32 * 16 * 8 * 2      # 4x8x16 bricks where W % 16 == 0
    assert L.call("pcrl_conv3d_k3_stats_rows", 32, 16, 16, 8, 128, 128, 1) == 32 * 4 * 1 * 1       # 4x8x16 bricks along (D, W, H) where H % 16 == 0
    assert L.call("pcrl_conv3d_k3_stats_rows", 32, 8, 8, 4, 256, 256, 1) == 32                     # 4x8x8 bricks (axis permutation)
    assert L.call("pcrl_conv3d_k3_fwd_kernel", 32, 16, 16, 8, 128, 128, 1) == 2 and L.call("pcrl_conv3d_k3_fwd_kernel", 32, 8, 8, 4, 256, 256, 1) == 1
    assert L.call("pcrl_conv3d_k3_fwd_kernel", 192, 4, 4, 4, 256, 256, 1) == 0 and L.call("pcrl_conv3d_k3_fwd_kernel", 32, 64, 64, 32, 64, 64, 0) == 0
    assert L.call("pcrl_upconv_dgrad_uses_brick", 32, 32, 32, 16, 128, 64, 1) == 1 and L.call("pcrl_upconv_dgrad_uses_brick", 32, 16, 16, 8, 256, 128, 1) == 1
    assert L.call("pcrl_upconv_dgrad_uses_brick", 32, 8, 8, 4, 512, 256, 1) == 1 and L.call("pcrl_upconv_dgrad_uses_brick", 32, 8, 8, 4, 512, 256, 0) == 0 and L.call("pcrl_upconv_fwd_uses_brick", 32, 16, 16, 8, 256, 128, 1) == 1
    assert L.call("pcrl_conv3d_k3_fwd_ws_bytes", 32, 64, 64, 32, 64, 64, 1) == 0


def test_error_path_reports_message():
    from pcrlv2_amd import _lib
    L = _lib.lib()
    with pytest.raises(_lib.PcrlError, match="multiples of 32"):
        L.call("pcrl_conv3d_k3_fwd", 16, 16, None, 16, None, 1, 4, 4, 4, 3, 32, 0, None)
    with pytest.raises(_lib.PcrlError, match="no CPU fallback"):
        L.call("pcrl_sigmoid_fwd", torch.zeros(4), torch.zeros(4), 4, None)


def test_model_api_and_state_dict(golden_dir):
    from pcrlv2_amd.models import PCRLv23d
    torch.manual_seed(0)
    m = PCRLv23d()
    keys = [l.split()[0] for l in open(os.path.join(golden_dir, "state_dict_manifest.txt"))]
    sd = m.state_dict()
    assert list(sd.keys()) == keys and len(keys) == 169
    assert sum(p.numel() for p in m.parameters()) == 17111434
    with pytest.raises(ValueError, match="normalization type xx is not supported"):
        PCRLv23d(norm="xx")
    with pytest.raises(ValueError, match="activation type silu is not supported"):
        PCRLv23d(act="silu")                                         # as the reference (:30); only the optional norm='gn' mode takes 'silu'
    assert "down_tr64.ops.0.bn1.running_mean" not in PCRLv23d(norm="gn", act="silu").state_dict()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 1, 16, 16, 16))
    m.set_compute_dtype("bf16")
    assert all(getattr(x, "compute_dtype", torch.bfloat16) == torch.bfloat16 for x in m.modules())
    # lazy num_batches_tracked
    m.down_tr64.ops[0]._count_batch()
    assert int(m.state_dict()["down_tr64.ops.0.bn1.num_batches_tracked"]) == 1


def test_lr_schedule_and_meter(golden_dir):
    from pcrlv2_amd.utils import AverageMeter, adjust_learning_rate
    lrs = np.load(os.path.join(golden_dir, "lr_schedule.npz"))["lr"]
    args = types.SimpleNamespace(lr=1e-3, epochs=240)
    opt = types.SimpleNamespace(param_groups=[dict(lr=0.0)])
    for e in (0, 1, 17, 120, 239, 240):
        adjust_learning_rate(e, args, opt)
        assert abs(opt.param_groups[0]["lr"] - lrs[e]) < 1e-18
    m = AverageMeter()
    m.update(2.0, 4)
    m.update(torch.tensor(4.0), 4)
    assert float(m.avg) == 3.0 and m.count == 8


def test_cos_loss_draw_order():
    """13 draws per step from python's global `random`, first draw selects index2 (train_3d.py:86-92,119-133)."""
    import random
    from pcrlv2_amd.train_3d import cos_loss

    class Cos:
        returns_mean = True

        def __call__(self, x, y):
            return (x * y).sum()
    f = [[torch.ones(2, 4), torch.ones(2, 4)] for _ in range(3)]
    random.seed(0)
    idx = [cos_loss(Cos(), f, f)[1] for _ in range(13)]
    random.seed(0)
    assert idx == [random.randint(0, 2) for _ in range(13)]
    loss, _ = cos_loss(Cos(), f, f)
    assert float(loss) == -8.0


def test_bucket_plan():
    from pcrlv2_amd.ddp import plan_buckets
    sizes = [10, 20, 5, 100, 7, 3]
    b = plan_buckets(sizes, 30)
    assert b[0][1] == sum(sizes) and b[-1][0] == 0
    assert all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))
    assert (35, 135) in b  # the oversize tensor gets its own bucket


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pcrlv2_amd.ddp import BucketedAllReduce, init_process_group_from_env
rank, world, _ = init_process_group_from_env("gloo")
sizes = [1000, 37, 5000, 12, 300]
flat = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1)
r = BucketedAllReduce(flat, sizes, bucket_mb=0.01)
assert len(r.buckets) >= 3
r.reduce()
exp = torch.arange(sum(sizes), dtype=torch.float32) * sum(range(1, world + 1))
assert torch.equal(flat, exp), (flat[:5], exp[:5])
dist.barrier()
print("OK", rank, flush=True)
dist.destroy_process_group()      # tear the group down before the interpreter exits (gloo threads alive at exit abort the process now and then)
'''


def test_bucketed_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("OK" in o for o in outs)


DP_WORKER = r'''
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pcrlv2_amd.ddp import DataParallel, init_process_group_from_env
from pcrlv2_amd import functions as Fn
rank, world, _ = init_process_group_from_env("gloo")
torch.manual_seed(0)

class Dot(torch.autograd.Function):
    # stand-in for a stage Function: parks its parameter gradient, reports it final when it ran in pass 0
    @staticmethod
    def forward(ctx, p, x, pass_idx):
        ctx.p, ctx.x, ctx.pass_idx = p, x, pass_idx
        return (p.detach() * x).sum()
    @staticmethod
    def backward(ctx, g):
        out = Fn._park(ctx.p, g * ctx.x)
        Fn.mark_final(ctx, [ctx.p])
        return out, None, None

shapes = [(300,), (7, 5), (2000,), (3,), (64, 8)]
params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
sizes = [p.numel() for p in params]
offs = [0]
for n in sizes: offs.append(offs[-1] + n)
flat_g = torch.full((offs[-1],), 123.0)                    # stale garbage that must not leak into the result
opt = types.SimpleNamespace(_plist=params, flat_g=flat_g, flat_p=torch.cat([p.detach().reshape(-1) for p in params]).clone(),
                            _gviews=[flat_g[o:o + n].view(p.shape) for p, o, n in zip(params, offs, sizes)], grad_scale=1.0, pre_step=None)
opt.gather_grads = lambda: None
model = torch.nn.Module()
dp = DataParallel(model, opt, bucket_mb=0.004, overlap=True)     # ~1000 floats per bucket -> several buckets
assert opt.grad_scale == 1.0 / world and len(dp.reducer.buckets) >= 3
for step in range(3):
    # parameter 3 has no gradient at all (unused head); parameter 1 gets a gradient only from a non-final pass
    x = [torch.full(s, float(rank + 1 + step)) for s in shapes]
    for p in params: p.grad = None
    # forward "pass 0" (final) for params 0,2,4, then "pass 1" for params 0,1,2,4; ONE backward replays pass 1 first
    l0 = sum(Dot.apply(params[i], x[i], 0) for i in (0, 2, 4))
    l1 = sum(Dot.apply(params[i], x[i], 1) for i in (0, 1, 2, 4))
    late = step == 2
    if late:      # a second backward after the buckets of 0,2,4 went out: the late-gradient path
        l0.backward(retain_graph=False)
        gone = [i for i in (0, 1, 2, 4) if dp._gathered[i]]
        assert len(gone) >= 1
        l1.backward()
        assert sorted(i for i, _ in dp._late) == gone, (dp._late, gone)
    else:
        (l0 + l1).backward()
        assert sum(dp._launched) >= 1 and not dp._late      # buckets went out from inside backward
    assert all(params[i].grad is not None for i in (0, 1, 2, 4)) and params[3].grad is None
    has = dp._pre_step(opt, None)
    assert has == [True, True, True, False, True], has
    tot = sum(r + 1 + step for r in range(world))
    for i, v in enumerate(opt._gviews):
        mult = {0: 2, 1: 1, 2: 2, 3: 0, 4: 2}[i]
        assert torch.allclose(v, torch.full_like(v, float(mult * tot))), (step, i, v.flatten()[:3], mult * tot)
dist.barrier()
print("OK", rank, flush=True)
dist.destroy_process_group()      # tear the group down before the interpreter exits (gloo threads alive at exit abort the process now and then)
'''


def test_data_parallel_overlap_logic_gloo_world2(tmp_path):
    """ddp.DataParallel (parked parameter gradients, bucket readiness via final-pass marks, zero-fill of grad-less parameters,
    sweep in step(), late gradients) with 2 gloo ranks on CPU and a stand-in optimizer."""
    script = tmp_path / "dp.py"
    script.write_text(DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("OK" in o for o in outs)


def test_model_init_matches_reference_under_seed():
    """Constructing PCRLv23d consumes the torch RNG in the reference's order: same seed => the reference's initial weights
    (golden from the real reference class, oracle/make_golden.py:make_init)."""
    import numpy as np
    import torch
    from pcrlv2_amd.models import PCRLv23d
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "init_seed7.npz"))
    torch.manual_seed(int(fx["meta/seed"]))
    sd = PCRLv23d().state_dict()
    assert len(sd) == (len(fx.files) - 1) // 3
    for k, v in sd.items():
        f = v.detach().double().reshape(-1).numpy()
        assert f.sum() == fx[k + "/sum"] and np.abs(f).sum() == fx[k + "/abs"], k
        assert np.array_equal(f[:4], fx[k + "/head"]), k


def test_model2d_api_and_state_dict():
    """PCRLv2 (2D) module tree: the attribute paths / key names the reference gets from smp.Unet('resnet18') + PCRLv2Decoder
    (pcrlv2_model.py:197-209; train_2d.py:99 saves model.model.encoder.state_dict(); README.md:40-44 loads it into a torchvision-named
    ResNet-18 encoder)."""
    from pcrlv2_amd.models import PCRLv2
    m = PCRLv2()
    enc = m.model.encoder.state_dict()
    assert len(enc) == 120 and not any(k.startswith("fc.") for k in enc)
    assert sum(p.numel() for p in m.model.encoder.parameters()) == 11176512          # torchvision resnet18 minus fc (513000)
    for k, shape in (("conv1.weight", (64, 3, 7, 7)), ("layer1.0.conv1.weight", (64, 64, 3, 3)), ("layer2.0.downsample.0.weight", (128, 64, 1, 1)),
                     ("layer4.1.bn2.running_var", (512,)), ("layer3.0.conv1.weight", (256, 128, 3, 3))):
        assert tuple(enc[k].shape) == shape, k
    sd = m.state_dict()
    for i, (cin, cout) in enumerate(((512, 256), (256, 128), (128, 64), (64, 32), (32, 16))):
        p = f"model.decoder.blocks.{i}."
        assert tuple(sd[p + "conv1.0.weight"].shape) == (cout, cin, 3, 3) and tuple(sd[p + "conv2.0.weight"].shape) == (cout, cout, 3, 3)
        assert (p + "conv1.0.bias") not in sd                                          # smp Conv2dReLU: bias=False with BatchNorm
        assert tuple(sd[p + "deep_supervision_head.0.bias"].shape) == (cout,) and tuple(sd[p + "deep_supervision_head.3.weight"].shape) == (3, cout, 1, 1)
        assert tuple(sd[p + "predictor_head.0.weight"].shape) == (2 * cout, cout) and tuple(sd[p + "predictor_head.3.weight"].shape) == (cout, 2 * cout)
        assert tuple(sd[p + "bn.weight"].shape) == (cout,)
    assert tuple(sd["model.segmentation_head.0.weight"].shape) == (3, 16, 3, 3)
    with __import__("pytest").raises(RuntimeError):
        m(__import__("torch").zeros(2, 3, 64, 64))                                     # CPU input: no fallback


def test_2d_encoder_checkpoint_round_trip_into_torchvision_named_resnet18(tmp_path):
    """What IS pinned of the 2D path without smp / torchvision in the image (the numerics stay "parity unpinned"): the on-disk
    contract.  A checkpoint of the layout train_2d.py:96-107 writes ('state_dict' = the ENCODER's state_dict) must load, after the
    README's `encoder_dict['fc.bias'] = 0; encoder_dict['fc.weight'] = 0` (README.md:40-44), into a ResNet-18 with torchvision's public
    key names the way smp's ResNetEncoder.load_state_dict does it (pop fc.*, strict load).  The skeleton below is torchvision's
    published BasicBlock ResNet-18 layout, written out; the file is produced by this engine's own model class."""
    import argparse
    import torch
    import torch.nn as nn
    from pcrlv2_amd.models import PCRLv2

    def block(cin, cout, stride):
        m = nn.Module()
        m.conv1, m.bn1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout)
        m.conv2, m.bn2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        if stride != 1 or cin != cout:
            m.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        return m

    class TorchvisionNamedResNet18(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
            for i, (cin, cout, stride) in enumerate(((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)), start=1):
                setattr(self, f"layer{i}", nn.Sequential(block(cin, cout, stride), block(cout, cout, 1)))

        def load_state_dict(self, state_dict, **kw):      # smp.encoders.resnet.ResNetEncoder.load_state_dict
            state_dict.pop("fc.bias", None)
            state_dict.pop("fc.weight", None)
            return super().load_state_dict(state_dict, **kw)

    torch.manual_seed(1)
    m = PCRLv2()
    args = argparse.Namespace(model="pcrlv2", n="chest", phase="pretask", ratio=0.8)
    path = tmp_path / "pcrlv2_chest_pretask_0.8_240.pt"
    torch.save({'opt': args, 'state_dict': m.model.encoder.state_dict(), 'optimizer': {}, 'epoch': 240}, path)
    encoder_dict = torch.load(path, weights_only=False)['state_dict']
    encoder_dict['fc.bias'] = 0
    encoder_dict['fc.weight'] = 0
    tv = TorchvisionNamedResNet18()
    res = tv.load_state_dict(encoder_dict, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(tv.state_dict().keys()) == list(m.model.encoder.state_dict().keys())          # same keys in the same order
    for k, v in tv.state_dict().items():
        assert torch.equal(v, m.model.encoder.state_dict()[k]), k
    # and back: the engine's --encoder_weights path takes a torchvision-named state_dict (with or without fc.*)
    sd = dict(tv.state_dict(), **{"fc.weight": torch.zeros(1000, 512), "fc.bias": torch.zeros(1000)})
    wpath = tmp_path / "resnet18.pth"
    torch.save(sd, wpath)
    m2 = PCRLv2(encoder_weights=str(wpath))
    for k, v in m2.model.encoder.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_c1_plumbing_2d_oracle_cpu():
    """BASELINE configs[0]: 2D PCRLv2 ResNet18-UNet, 224x224 crops, b=4, CPU-only torch -- plumbing (shapes, finite, loss goes down)
    on the CPU oracle restatement (oracle/pcrlv2_2d_oracle.py, PARITY UNPINNED), initial weights from the engine's model class."""
    import random
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd.models import PCRLv2
    torch.manual_seed(0)
    m = PCRLv2()
    pn = [n for n, _ in m.named_parameters()]
    sd = {k: (v.detach().clone().requires_grad_(k in pn) if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    batch = O.synthetic_batch(4, 224, 96, seed=0)
    random.seed(0)
    losses = []
    for step in range(3):
        so = {}
        r = O.step_losses(sd, batch, epoch=0, so=so)
        if step == 0:
            assert r["mask1"].shape == (4, 3, 224, 224) and all(t.shape == (4, 3, 224, 224) for t in r["mid1"])
            assert [tuple(p.shape) for p, _ in r["out1"]] == [(4, c) for c in (256, 128, 64, 32, 16)]
        assert torch.isfinite(r["loss"])
        losses.append(float(r["loss1"]))
        r["loss"].backward()
        with torch.no_grad():
            for k in pn:
                if sd[k].grad is not None:
                    sd[k] -= 0.05 * sd[k].grad
                    sd[k].grad = None
            for k, v in so.items():
                sd[k] = v
    assert losses[-1] < losses[0], losses


def test_composed_upconv_algebra_in_float64():
    """The identity csrc/upconv_fused.hip builds on, restated in torch float64 on the CPU (no library call): ConvTranspose3d(k2,s2) followed
    by Conv3d(3x3x3, pad 1) equals, phase by phase, an 8-tap operator on the zero-padded coarse tensor with
    Weff[p][q] = sum over the (t, s) pairs of (p, q) of Wup[:, :, s] W0[:, :, t]^T   (per axis: (0,0): (0,1); (0,1): (1,0),(2,1); (1,0): (0,0),(1,1); (1,1): (2,0))
    plus a bias that depends on the border class of the fine voxel (the inner convolution zero-pads the UPSAMPLED tensor)."""
    import itertools
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    N, D, H, W, Ci, Cm, Co = 2, 2, 3, 2, 3, 4, 5
    x = torch.randn(N, Ci, D, H, W, generator=g, dtype=torch.float64)
    wup, bup = torch.randn(Ci, Cm, 2, 2, 2, generator=g, dtype=torch.float64), torch.randn(Cm, generator=g, dtype=torch.float64)
    w0, b0 = torch.randn(Co, Cm, 3, 3, 3, generator=g, dtype=torch.float64), torch.randn(Co, generator=g, dtype=torch.float64)
    ref = F.conv3d(F.conv_transpose3d(x, wup, bup, stride=2), w0, b0, padding=1)
    pairs = {(0, 0): [(0, 1)], (0, 1): [(1, 0), (2, 1)], (1, 0): [(0, 0), (1, 1)], (1, 1): [(2, 0)]}
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))                       # zero padding of the COARSE tensor
    out = torch.zeros_like(ref)
    for p in itertools.product((0, 1), repeat=3):
        acc = torch.zeros(N, Co, D, H, W, dtype=torch.float64)
        for q in itertools.product((0, 1), repeat=3):
            weff = torch.zeros(Ci, Co, dtype=torch.float64)
            for (td, sd), (th, sh), (tw, sw) in itertools.product(pairs[p[0], q[0]], pairs[p[1], q[1]], pairs[p[2], q[2]]):
                weff += wup[:, :, sd, sh, sw] @ w0[:, :, td, th, tw].t()
            # coarse voxel v + p - 1 + q  ->  index v + p + q in the padded tensor
            sl = xp[:, :, p[0] + q[0]:p[0] + q[0] + D, p[1] + q[1]:p[1] + q[1] + H, p[2] + q[2]:p[2] + q[2] + W]
            acc += torch.einsum("ncdhw,co->nodhw", sl, weff)
        out[:, :, p[0]::2, p[1]::2, p[2]::2] = acc
    # bias: b0 + the taps of w0 that stay inside the fine grid applied to b_up
    FD, FH, FW = 2 * D, 2 * H, 2 * W
    ok = lambda t, f, n: 0 <= f + t - 1 < n
    for fd, fh, fw in itertools.product(range(FD), range(FH), range(FW)):
        bias = b0.clone()
        for td, th, tw in itertools.product(range(3), repeat=3):
            if ok(td, fd, FD) and ok(th, fh, FH) and ok(tw, fw, FW):
                bias += w0[:, :, td, th, tw] @ bup
        out[:, :, fd, fh, fw] += bias
    assert torch.allclose(out, ref, rtol=1e-12, atol=1e-12), float((out - ref).abs().max())


GUARD_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pcrlv2_amd.ddp import init_process_group_from_env
from pcrlv2_amd.train_3d import divergence_flag
rank, world, _ = init_process_group_from_env("gloo")
# only rank 1 diverges (loss > 1000): the decision is collective, both ranks hold flag 1 and would skip together
f = divergence_flag(torch.tensor(2.5e3 if rank == 1 else 0.7))
assert f.shape == (1,) and float(f) == 1.0, (rank, f)
# nobody diverges: flag 0 on both; exactly 1000 is not "> 1000" (train_3d.py:140)
assert float(divergence_flag(torch.tensor(1000.0))) == 0.0
# NaN > 1000 is False in the reference as well: a NaN loss does not trip the guard
assert float(divergence_flag(torch.tensor(float("nan")))) == 0.0
dist.barrier()
print("OK", rank, flush=True)
dist.destroy_process_group()      # tear the group down before the interpreter exits (gloo threads alive at exit abort the process now and then)
'''


def test_divergence_guard_decision_is_collective_gloo_world2(tmp_path):
    """train_3d.py:140-142 under one process per GPU: a rank that skipped alone would leave its peers in the gradient all-reduce."""
    script = tmp_path / "g.py"
    script.write_text(GUARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29735", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("OK" in o for o in outs)


def test_bench_gpus4_code_path_dry_run_gloo_world4():
    """`python bench.py --gpus 4` end to end on CPU (PCRL_BENCH_DRYRUN=1): the self-spawn through torch.distributed.run, four gloo
    ranks, the rank / device table, the four-setting bucket A/B, the barrier-bracketed timed region, the JSON line -- with the step replaced by a
    stub whose KNOWN gradients go through the engine's parking / bucket / all-reduce machinery and are checked after every optimizer step."""
    import json
    env = dict(os.environ, PCRL_BENCH_DRYRUN="1", PCRL_BIND_CPUS="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines and lines[-1].startswith("{"), r.stdout[-500:]        # the JSON line is the LAST line of stdout
    assert sum(ln.startswith("{") for ln in lines) == 1                 # ... and the only one
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 4 and d["config"]["parallelism"] == "dp4" and d["scaling"] == "weak"
    ranks = d["distributed"]["ranks"]
    assert sorted(x["rank"] for x in ranks) == [0, 1, 2, 3] and len({x["device"] for x in ranks}) == 4
    ab = d["distributed"]["ab"]
    assert set(ab["settings"]) == {"from_step_buckets24MB", "overlap_buckets24MB", "from_step_1bucket", "overlap_1bucket"}
    assert ab["used_for_timed_region"] in ab["settings"] and ab["settings"]["overlap_buckets24MB"]["buckets"] >= 3
    assert abs(d["per_gpu_value"] * 4 - d["value"]) < 0.02 * d["value"]
    assert d["dry_run"]["gradient_checks_passed"] >= 4 + 1 + 4 * 8 and "DRY RUN" in d["config"]["workload"]
    assert d["steps"] == 4 and d["warmup"] == 1 and d["ms_per_step"] > 0


def test_bnr_counted_wait_matches_the_stores_behind_the_dma_pieces(tmp_path):
    """the fused data-gradient + BatchNorm-reduce epilogue (csrc/conv_brick16.h, BNR instantiations) knows that its LDS-DMA pieces have
    landed from a COUNTED wait -- `s_waitcnt vmcnt(12 FN)`: the pieces are the oldest vector-memory operations in flight and exactly 12 FN stores (three
    lines of the output tile) were issued behind them.  If a compiler ever merges, drops or reorders those stores the count is wrong and the partial
    sums read a half-landed tile.  Build-time check on the ISA hipcc emits for every BNR instantiation (gfx950 cross-compile, ~5 s): between the last
    `global_load_lds` request and the counted wait there are exactly N vector-memory instructions, all stores, N = the wait's count, straight-line."""
    import re
    import shutil
    from pcrlv2_amd import build as B
    hipcc = B.hipcc()
    src = os.path.join(ROOT, "pcrlv2_amd", "csrc", "conv_brick16_bnr.hip")
    out = tmp_path / "bnr.s"
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)] + B.FILE_FLAGS.get("conv_brick16_bnr.hip", [])
    r = subprocess.run([hipcc, *flags, "--cuda-device-only", "-S", src, "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().split("\n")
    waits = [i for i, ln in enumerate(lines) if re.search(r"s_waitcnt vmcnt\((48|24)\)", ln)]
    assert len(waits) == 8, ("one counted wait per BNR instantiation (2 widths x 2 axis orders x 3D / 2D)", len(waits))
    for w in waits:
        want = int(re.search(r"vmcnt\((\d+)\)", lines[w]).group(1))
        j, vm = w - 1, []
        while j > 0 and "global_load_lds" not in lines[j]:
            ln = lines[j].strip()
            assert not (ln.endswith(":") and not ln.startswith(";")), ("a label (control flow) between the DMA requests and the counted wait", w, ln)
            if re.match(r"(global|buffer|flat|scratch)_", ln):
                vm.append(ln.split()[0])
            j -= 1
        assert j > 0, "no LDS-DMA request in front of the counted wait"
        assert len(vm) == want and all(v.startswith("global_store") for v in vm), (w, want, len(vm), sorted(set(vm)))


class _BusyLoops:
    """`n` busy-loop processes beside a test (the rc = 1 exits of the N-rank entry points showed up under CPU load, when a
    rank's teardown was slow enough for its peers' threads to outlive it).  Killed by their exact PIDs."""

    def __init__(self, n):
        self.n, self.procs = n, []

    def __enter__(self):
        self.procs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(self.n)]
        return self

    def __exit__(self, *exc):
        for p in self.procs:
            p.kill()
        for p in self.procs:
            p.wait()


_ABORT_MARKS = ("terminate called", "destroy_process_group() was not called", "ChildFailedError", "Aborted", "SIGABRT")


def test_bench_gpus4_dry_run_exits_clean_10_times_under_cpu_load():
    """Every rank of `bench.py --gpus N` leaves through ddp.shutdown() (barrier + destroy_process_group).  Round 5's bench.py returned from main()
    with the group alive: one dry run in four ended `terminate called without an active exception` on some rank AFTER the JSON line and torchrun
    reported rc = 1 -- on the day the driver runs `--gpus 8` that voids the record.  Ten spawns with eight busy loops beside them: rc = 0 and no
    abort / missing-teardown message every time."""
    env = dict(os.environ, PCRL_BENCH_DRYRUN="1", PCRL_BIND_CPUS="0")
    env.pop("WORLD_SIZE", None)
    with _BusyLoops(8):
        for i in range(10):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert r.returncode == 0, (i, r.stderr[-3000:])
            assert not any(m in r.stderr for m in _ABORT_MARKS), (i, r.stderr[-3000:])
            assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1, (i, r.stdout[-300:])


MAIN_STUB_WORKER = r'''
# `main.py --data synthetic --d 3` on CPU, two gloo ranks: main.main -> train_3d.train_pcrlv2_3d -> (group created) -> epochs -> ddp.shutdown.
# The model / optimizer / step are stand-ins (no GPU here); everything that decides how the process ENDS is the product's own code.
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from pcrlv2_amd import main as M, train_3d as T, ddp, config
torch.cuda.set_device = lambda *_a, **_k: None
config.EMPTY_CACHE_PER_EPOCH = False


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1000))

    def cuda(self, *a, **k):
        return self

    def set_compute_dtype(self, dt):
        return self


class Opt(torch.optim.SGD):
    def __init__(self, params, lr, momentum, weight_decay):
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay)


class Crit(torch.nn.Module):
    def cuda(self):
        return self


def dp(model, opt):
    assert dist.is_initialized() and dist.get_world_size() == 2
    opt.collective = True


def step(model, opt, batch, epoch, crit, cosine, guard=True):
    g = torch.full((1000,), float(dist.get_rank() + 1))
    dist.all_reduce(g)                       # the step's exchange
    assert float(g[0]) == 3.0
    z = torch.zeros(())
    out = T.StepLosses((z, z, z, z, z)) if hasattr(T, "StepLosses") else None
    return out


T.PCRLv23d, T.FusedSGD, T.MSELoss, T.CosineSimilarityMean = Net, Opt, Crit, Crit
T._ddp.DataParallel = dp
T.train_pcrlv2_inner = lambda args, epoch, loader, model, opt, crit, cos, verbose=True: [step(model, opt, b, epoch, crit, cos) for b in loader]
M.SyntheticLunaLoader.__init__ = lambda self, b, steps, seed=0, device=None: setattr(self, "steps", steps)
M.SyntheticLunaLoader.__iter__ = lambda self: iter(range(self.steps))
mode = sys.argv[3]
if mode == "raise" and os.environ["RANK"] == "1":
    def boom(*a, **k):
        raise RuntimeError("rank 1 fails inside the epoch")
    T.train_pcrlv2_inner = boom
try:
    M.main(["--data", "synthetic", "--d", "3", "--b", "4", "--epochs", "1", "--steps_per_epoch", "3", "--output", sys.argv[2], "--gpus", "0,1"])
except RuntimeError as e:
    assert mode == "raise" and "rank 1 fails" in str(e), e
    assert not dist.is_initialized(), "the group must be torn down when an exception propagates"
    print("OK-raised", flush=True)
    raise SystemExit(0)
assert not dist.is_initialized(), "train_pcrlv2_3d created the group and must have destroyed it"
print("OK", os.environ["RANK"], flush=True)
'''


def test_main_py_two_ranks_cpu_stub_exits_clean_10_times_under_cpu_load(tmp_path):
    """`main.py --gpus 0,1` (one process per GPU instead of train_3d.py:54's nn.DataParallel) tears its process group down in a `finally`
    (train_3d.train_pcrlv2_3d -> ddp.shutdown): ten 2-rank gloo runs of main.main with stand-in model / step on CPU, eight busy loops beside
    them, every rank rc = 0 with the group gone; and once with rank 1 raising inside the epoch -- its group is destroyed on the way out
    (rank 0, left alone in the all-reduce, is killed by the test: only rank 1's exit is asserted there)."""
    script = tmp_path / "m.py"
    script.write_text(MAIN_STUB_WORKER)

    def spawn(port, mode):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", PCRL_BIND_CPUS="0")
        return [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), mode], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    with _BusyLoops(8):
        for i in range(10):
            procs = spawn(29741 + i, "ok")
            outs = [p.communicate(timeout=180)[0] for p in procs]
            assert all(p.returncode == 0 for p in procs), (i, outs)
            assert all("OK" in o for o in outs) and not any(m in o for o in outs for m in _ABORT_MARKS), (i, outs)
    procs = spawn(29761, "raise")
    out1 = procs[1].communicate(timeout=180)[0]
    procs[0].kill()
    procs[0].communicate()
    assert procs[1].returncode == 0 and "OK-raised" in out1 and not any(m in out1 for m in _ABORT_MARKS), out1


@pytest.mark.parametrize("tag", ["loss2d_b4_5scales", "loss2d_b2_nl3"])
def test_2d_loss_assembly_and_draw_order_match_the_reference(tag, golden_dir, monkeypatch):
    """What can be pinned of the 2D path without smp / torchvision: pcrlv2_amd.train_2d's `cos_loss` and loss assembly against
    fixtures the REFERENCE's own train_2d.cos_loss produced (oracle/make_golden.py --loss2d: train_2d.py:111-117 imported, :139-168 restated around
    it) on closed-form five-scale feature lists: all five losses to float64 round-off, the deep-supervision scale, and the 1 + 2 * nlocal scale draws
    in the reference's order.  The 2D MODEL remains 'parity unpinned'."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcrlv2_2d_oracle as O2
    from pcrlv2_amd import train_2d, train_3d
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    b, nl, epoch = int(fx["b"]), int(fx["nlocal"]), int(fx["epoch"])
    f1, f2, fl, mask1, masks1, gt = O2.fill_loss_inputs(b, nl, dtype=torch.float64)
    draws = []
    real = random.randint
    monkeypatch.setattr(train_3d.random, "randint", lambda a_, b_: (draws.append(real(a_, b_)) or draws[-1]))
    random.seed(int(fx["seed"]))
    total, l1, l2, l4, ll = train_2d.assemble_losses(f1, f2, fl, mask1, masks1, gt, b, nl, epoch, torch.nn.MSELoss(), torch.nn.CosineSimilarity())
    assert draws == [int(v) for v in fx["draws"]] and draws[0] == int(fx["index2"])
    for name, got in (("loss", total), ("loss1", l1), ("loss2", l2), ("loss4", l4), ("local_loss", ll)):
        assert abs(float(got) - float(fx[name])) < 1e-12, (name, float(got), float(fx[name]))
    assert train_2d.cos_loss is train_3d.cos_loss       # one implementation serves both loops (train_2d.py:111-117 == train_3d.py:86-92)


def test_rank_cpu_binding_helpers():
    """ddp.bind_rank_to_numa's pure parts: sysfs cpulist parsing and the even split of a NUMA node's CPUs among the ranks on it."""
    from pcrlv2_amd.ddp import cpu_share, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("5") == [5]
    node = list(range(0, 64)) + list(range(128, 192))           # one socket of a 2 x 64-core SMT box
    allowed = set(range(256))
    shares = [cpu_share(node, allowed, 4, k) for k in range(4)]
    assert all(len(s) == 32 for s in shares) and len(set(c for s in shares for c in s)) == 128      # disjoint, whole node
    assert cpu_share(node, {1, 2, 3}, 4, 3) == [1, 2, 3]        # fewer CPUs than ranks: share the pool rather than starve a rank
    assert cpu_share([], {4, 5, 6, 7}, 2, 1) == [6, 7]          # unknown node: even split of what is allowed
    # the rank's share is split again: the launcher thread keeps CPUs of its own, the loader workers get the rest
    from pcrlv2_amd.ddp import split_share
    launcher, workers = split_share(shares[1], 8)
    assert len(launcher) == 24 and len(workers) == 8 and not set(launcher) & set(workers) and sorted(launcher + workers) == shares[1]
    assert split_share(shares[0], 0) == (shares[0], shares[0])                    # no workers announced: nothing to split
    small = list(range(6))
    assert split_share(small, 8) == (small, small)                              # share too small: not split (everything time-slices, and the log says so)
    la, wo = split_share(list(range(10)), 8)
    assert la == [0, 1] and wo == list(range(2, 10))                            # the launcher never gets fewer than two CPUs


def test_loader_worker_moves_to_the_cpus_the_binding_left_for_it(monkeypatch):
    """data.worker_affinity_init: a forked worker leaves the launcher's mask for $PCRL_WORKER_CPUS."""
    from pcrlv2_amd import data
    if not hasattr(os, "sched_setaffinity"):
        return
    allowed = sorted(os.sched_getaffinity(0))
    try:
        monkeypatch.setenv("PCRL_WORKER_CPUS", ",".join(str(c) for c in allowed[-2:]))
        data.worker_affinity_init(0)
        assert sorted(os.sched_getaffinity(0)) == allowed[-2:]                  # outside a DataLoader (no worker info): the whole list
        monkeypatch.setenv("PCRL_WORKER_CPUS", "")
        data.worker_affinity_init(0)                                            # no variable: nothing happens
        assert sorted(os.sched_getaffinity(0)) == allowed[-2:]
        # one CPU per worker only where the rank's share was really split: inside a DataLoader worker (worker info present) ...
        import types
        monkeypatch.setattr(torch.utils.data, "get_worker_info", lambda: types.SimpleNamespace(num_workers=2, id=1))
        monkeypatch.setenv("PCRL_WORKER_CPUS", ",".join(str(c) for c in allowed[-2:]))
        monkeypatch.setenv("PCRL_WORKER_CPUS_SPLIT", "0")                       # ... an UNSPLIT share is shared: the workers float over all of it
        os.sched_setaffinity(0, allowed)
        data.worker_affinity_init(1)
        assert sorted(os.sched_getaffinity(0)) == allowed[-2:]
        monkeypatch.setenv("PCRL_WORKER_CPUS_SPLIT", "1")                       # ... a split share: worker i on CPU i of the list
        os.sched_setaffinity(0, allowed)
        data.worker_affinity_init(1)
        assert sorted(os.sched_getaffinity(0)) == [allowed[-1]]
    finally:
        os.sched_setaffinity(0, allowed)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: include/pcrl_hip.h must compile as plain C (gcc -std=c99 -pedantic, no C++, no HIP headers), and a C
    program linked against libpcrl_hip.so must be able to call it -- here the host-only entry points (version string, workspace-size and
    dispatch queries, an argument error with its message); compute calls need a GPU and live in the -m gpu tests."""
    import shutil
    import subprocess
    from pcrlv2_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    if not os.path.exists(_lib.LIBPATH):
        import __graft_entry__ as g
        g.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "pcrl_hip.h"
int main(void) {
  const char* v = pcrl_version();
  if (!v || !strstr(v, "gfx950")) return 1;
  if (pcrl_conv3d_k3_wgrad_ws_bytes(2, 16, 16, 16, 64, 64) <= 0) return 2;
  if (pcrl_bn_bwd_partial_rows((int64_t)1 << 22) != 4096) return 3;
  if (pcrl_prelu_bwd_partial_rows(1000) != 4) return 4;
  /* an argument error: negative code, message available, nothing thrown across the boundary */
  int rc = pcrl_prelu_fwd(NULL, NULL, NULL, 16, 3, PCRL_BF16, NULL);
  if (rc != PCRL_EINVAL || !strlen(pcrl_last_error())) return 5;
  printf("%s\n", v);
  return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIBPATH)
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lpcrl_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "gfx950" in r.stdout, (r.returncode, r.stdout, r.stderr)
