"""2D path: PCRLv2 (ResNet-18 U-Net) forward / losses / gradients / SGD on the HIP engine against the CPU oracle
(oracle/pcrlv2_2d_oracle.py, PARITY UNPINNED: a restatement, the reference's 2D model cannot be imported in this image)."""
import math
import os
import random
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))


def _build(seed=3, dtype=torch.float32):
    from pcrlv2_amd.models import PCRLv2
    torch.manual_seed(seed)
    model = PCRLv2().cuda().set_compute_dtype(dtype)
    # non-trivial affine parameters / biases so that every gradient path is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_((torch.rand(p.shape, generator=g) * 0.5 + (0.75 if n.endswith("weight") else -0.25)).to(p.device))
    return model


def _oracle_state(model):
    pn = set(n for n, _ in model.named_parameters())
    sd = {}
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            t = v.detach().cpu().double()
            if k in pn:
                t.requires_grad_(True)
            sd[k] = t
        else:
            sd[k] = v.cpu()
    return sd


@pytest.mark.parametrize("b,size,local", [(4, 64, 32), (4, 224, 96)])     # the second is BASELINE configs[0] (C1: 224x224, b=4) on the GPU
def test_step_matches_oracle_fp32(b, size, local):
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    model = _build()
    sd = _oracle_state(model)
    batch = O.synthetic_batch(b, size, local, seed=11)
    random.seed(5)
    so = {}
    ref = O.step_losses(sd, tuple(t.double() if torch.is_tensor(t) else [u.double() for u in t] for t in batch), epoch=3, so=so)
    ref["loss"].backward()
    random.seed(5)
    model.train()
    model.zero_grad(set_to_none=True)
    got = train_2d.step_losses(model, batch, 3, train_2d.MSELoss2d(), CosineSimilarityMean())
    got[0].backward()
    for name, g_, r_ in (("loss", got[0], ref["loss"]), ("loss1", got[1], ref["loss1"]), ("loss2", got[2], ref["loss2"]),
                         ("loss4", got[3], ref["loss4"]), ("local", got[4], ref["local_loss"])):
        assert abs(float(g_) - float(r_)) < 2e-4 * max(1.0, abs(float(r_))), (name, float(g_), float(r_))
    worst = []
    for name, p in model.named_parameters():
        r = sd[name].grad
        if r is None:
            assert p.grad is None, f"{name}: the oracle has no gradient, the HIP path produced one"
            continue
        assert p.grad is not None, f"{name}: missing gradient"
        rn = float(r.norm())
        if rn < 1e-9:      # conv biases in front of BatchNorm: identically zero
            assert float(p.grad.abs().max()) < 1e-6, name
            continue
        worst.append((float((p.grad.double().cpu() - r).norm()) / rn, name))
    worst.sort(reverse=True)
    # Tolerance: float32 round-off is amplified by the cancellation inside every BatchNorm backward on the way down (the heads and the
    # last decoder block agree to 1e-5..1e-4, the error grows layer by layer).  The same step in plain PyTorch-CPU float32 against its
    # own float64 run shows median 5.4e-3 / max 7.4e-3 at b=8 (tests/probe2d_conditioning.py prints ours: 2.8e-3 / 4.9e-3).
    assert worst[0][0] < 3e-2, worst[:5]
    assert sorted(w for w, _ in worst)[len(worst) // 2] < 1e-2, worst[:5]
    heads = [w for w, n in worst if "predictor_head" in n or "segmentation_head" in n or n.endswith(".bn.weight")]
    assert max(heads) < 5e-4, sorted(heads)[-3:]
    # running statistics of every BatchNorm after the three forwards of the step are chained updates; check the last written ones
    model.flush_counters()
    msd = model.state_dict()
    assert int(msd["model.encoder.bn1.num_batches_tracked"]) == 3


def test_forward_shapes_and_outputs_fp32():
    import pcrlv2_2d_oracle as O
    model = _build(seed=9)
    sd = _oracle_state(model)
    x = O.synthetic_batch(3, 96, 32, seed=2)[0]
    model.train()
    outs, masks, mids = model(x.cuda())
    so = {}
    r_outs, r_masks, r_mids = O.model_forward(x.double(), sd, so=so)
    assert masks.shape == (3, 3, 96, 96) and len(mids) == 5 and all(m.shape == (3, 3, 96, 96) for m in mids)
    assert [tuple(p.shape) for p, _ in outs] == [(3, c) for c in (256, 128, 64, 32, 16)]

    def rel(a, b):
        return float((a.detach().double().cpu() - b.detach()).norm() / b.detach().norm())
    assert rel(masks, r_masks) < 1e-4
    for a, b in zip(mids, r_mids):
        assert rel(a, b) < 1e-4
    for (p, q), (rp, rq) in zip(outs, r_outs):
        assert rel(p, rp) < 1e-4 and rel(q, rq) < 1e-4
    # running statistics
    model.flush_counters()
    msd = model.state_dict()
    for k, v in so.items():
        assert rel(msd[k], v) < 1e-4, k
    # local=True: no segmentation map (pcrlv2_model.py:206-208)
    outs_l, masks_l, mids_l = model(x.cuda(), local=True)
    assert masks_l is None and len(mids_l) == 5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_steps_reduce_loss(dtype):
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    model = _build(seed=1, dtype=dtype)
    opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    batch = O.synthetic_batch(4, 64, 32, seed=4)
    random.seed(0)
    crit, cos = train_2d.MSELoss2d(), CosineSimilarityMean()
    losses = []
    for _ in range(8):
        out = train_2d.train_step(model, opt, batch, 0, crit, cos)
        losses.append(float(out[1]))           # restoration term: deterministic target, must go down on a repeated batch
    assert all(l == l for l in losses), losses
    assert losses[-1] < losses[0], losses
    n_none = sum(p.grad is None for p in model.parameters())
    assert n_none >= 1    # view-2 / unselected deep-supervision heads get no gradient in a step (autograd semantics of the reference)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_second_view_on_its_own_stream_is_bit_identical_2d(dtype):
    """config.VIEW_STREAMS_2D (default on): the second global view's forward and backward run on their own stream next to the first's; the
    packed-weight caches and the BatchNorm running statistics are ordered by events.  Same kernels, same operands -> parameters, momentum
    buffers and running statistics after three steps are BIT-identical to the one-stream run."""
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import config, train_2d
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    batches = [O.synthetic_batch(4, 64, 32, seed=11 + k) for k in range(3)]
    keep, finals = config.VIEW_STREAMS_2D, []
    try:
        for on in (True, False):
            config.VIEW_STREAMS_2D = on
            model = _build(seed=5, dtype=dtype)
            opt = FusedSGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
            random.seed(2)
            for bt in batches:
                out = train_2d.train_step(model, opt, bt, 0, train_2d.MSELoss2d(), CosineSimilarityMean())
            torch.cuda.synchronize()
            rs = torch.cat([v.flatten().float() for k, v in sorted(model.state_dict().items()) if "running" in k])
            finals.append(([float(o) for o in out], opt.flat_p.clone(), opt.flat_buf.clone(), rs))
    finally:
        config.VIEW_STREAMS_2D = keep
    a, b = finals
    assert a[0] == b[0], (a[0], b[0])
    for x, y, what in zip(a[1:], b[1:], ("parameters", "momentum buffers", "running statistics")):
        assert torch.equal(x, y), what


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_as_one_autograd_node_is_bit_identical_2d(dtype):
    """functions2d.EncoderFn (PCRL_FUSED_ENCODER_2D, default on): BatchNorm apply + identity add + ReLU in one pass, the two gradients of a block's
    output summed inside the mask / max-pool backward passes, the stem's apply + ReLU + MaxPool2d in one pass -- against one autograd node per
    unit with autograd's own adds: parameters, momentum buffers and running statistics after three steps are BIT-identical."""
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import functions2d as Fn2, train_2d
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    from pcrlv2_amd import config
    batches = [O.synthetic_batch(4, 64, 32, seed=21 + k) for k in range(3)]
    keep, keep_stem, keep_bnred, finals = Fn2.FUSED_ENCODER, Fn2.STEM_KERNEL, config.DGRAD_BNRED, []
    try:
        Fn2.STEM_KERNEL = False      # the dedicated stem kernels (another summation order) exist on the one-node path only: compared separately below
        # likewise the data gradient that takes bn1's first backward pass with it (another summation order of two per-channel sums; it knows a BasicBlock's
        # structure, so it lives on the one-node path only): held to the separate pass in tests/test_dgrad_bnred_gpu.py
        config.DGRAD_BNRED = False
        for on in (True, False):
            Fn2.FUSED_ENCODER = on
            model = _build(seed=6, dtype=dtype)
            opt = FusedSGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
            random.seed(4)
            for bt in batches:
                out = train_2d.train_step(model, opt, bt, 0, train_2d.MSELoss2d(), CosineSimilarityMean())
            torch.cuda.synchronize()
            sd = model.state_dict()
            rs = torch.cat([v.flatten().float() for k, v in sorted(sd.items()) if "running" in k or "num_batches" in k])
            finals.append(([float(o) for o in out], opt.flat_p.clone(), opt.flat_buf.clone(), rs))
    finally:
        Fn2.FUSED_ENCODER, Fn2.STEM_KERNEL, config.DGRAD_BNRED = keep, keep_stem, keep_bnred
    a, b = finals
    assert a[0] == b[0], (a[0], b[0])
    for x, y, what in zip(a[1:], b[1:], ("parameters", "momentum buffers", "running statistics")):
        assert torch.equal(x, y), what


def test_stem_kernels_in_the_step_2d():
    """PCRL_STEM_KERNEL_2D (default on; bf16, H/2 % 8 == 0 and W/2 % 32 == 0): the dedicated 7x7 kernels on the float32 NCHW image against the
    general gather kernel on the image padded to 8 channels -- same bf16-rounded operands, another summation order: a sanity check at step
    level (bf16 noise bounds); the kernels themselves are held to float64 torch in tests/test_ops2d_gpu.py::test_stem_kernels_against_torch."""
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import functions2d as Fn2, train_2d
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    batch = O.synthetic_batch(4, 64, 32, seed=41)
    keep, res = Fn2.STEM_KERNEL, []
    try:
        for on in (True, False):
            Fn2.STEM_KERNEL = on
            model = _build(seed=9, dtype=torch.bfloat16)
            model.train()
            random.seed(3)
            losses = train_2d.step_losses(model, batch, 3, train_2d.MSELoss2d(), CosineSimilarityMean())
            losses[0].backward()
            torch.cuda.synchronize()
            res.append(([float(l) for l in losses], {n: (None if p.grad is None else p.grad.double().cpu()) for n, p in model.named_parameters()}))
    finally:
        Fn2.STEM_KERNEL = keep
    (la, ga), (lb, gb) = res
    # a different float32 summation order flips the bf16 rounding of a few stem outputs; 30 batch-statistics layers over b = 4 tiny images carry
    # that to 1e-3-level loss differences (measured 3.1e-3 on the total) -- the same size as bf16 against float32 on this fixture
    for x, y in zip(la, lb):
        assert abs(x - y) < 8e-3 * max(1.0, abs(y)), (la, lb)
    # gradients: the None pattern only -- on this b = 4 fixture the gradients of the full loss are chaotic under ANY change of bf16 rounding
    # (measured median 0.32 rel-L2 between the two kernels; tests/test_model_gpu.py's header documents the same for the 3D fixture)
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_step_equals_the_reference_shaped_step_2d(dtype):
    """train_2d.step_losses' engine form (scales drawn before the forwards, unread stateless outputs skipped, one launch for the 26 cosine
    means, fused restoration terms; PCRL_FUSED_STEP_2D) against the reference-shaped loop over the public model API with the same draws:
    the same five losses, the same gradients (float32: 1e-5 relative per tensor; bf16: 5e-2 -- the deep-supervision gradient takes a
    float32 instead of a bf16 route into the 1x1 convolution's backward), the same None pattern, identical running statistics and
    num_batches_tracked (the skipped work has no state)."""
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    batch = O.synthetic_batch(4, 64, 32, seed=31)
    res = []
    keep = train_2d.FUSED_STEP_2D
    try:
        for on in (True, False):
            train_2d.FUSED_STEP_2D = on
            model = _build(seed=8, dtype=dtype)
            model.train()
            random.seed(9)
            losses = train_2d.step_losses(model, batch, 3, train_2d.MSELoss2d(), CosineSimilarityMean())
            after = random.random()            # the global stream advanced by exactly the same 13 draws
            losses[0].backward()
            torch.cuda.synchronize()
            sd = model.state_dict()
            res.append(([float(l) for l in losses], {n: (None if p.grad is None else p.grad.double().cpu()) for n, p in model.named_parameters()},
                        {k: v.clone() for k, v in sd.items() if "running" in k or "num_batches" in k}, after))
    finally:
        train_2d.FUSED_STEP_2D = keep
    (la, ga, ba, ra), (lb, gb, bb, rb) = res
    assert ra == rb
    ltol = 1e-5 if dtype == torch.float32 else 2e-3
    for x, y in zip(la, lb):
        assert abs(x - y) < ltol * max(1.0, abs(y)), (la, lb)
    gtol = 1e-5 if dtype == torch.float32 else 5e-2      # bf16: rounding noise (measured up to 2.1e-2 on the stem's BatchNorm bias); the float32 run is the equivalence check
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n
        if ga[n] is not None and float(gb[n].norm()) > 1e-6:      # (biases in front of a BatchNorm: identically zero, round-off either way)
            assert float((ga[n] - gb[n]).norm()) <= gtol * float(gb[n].norm()), (n, float((ga[n] - gb[n]).norm()) / float(gb[n].norm()))
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k


def test_bf16_step_close_to_fp32():
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    batch = O.synthetic_batch(4, 64, 32, seed=7)
    vals = {}
    for dt in (torch.float32, torch.bfloat16):
        model = _build(seed=2, dtype=dt)
        random.seed(1)
        vals[dt] = [float(v) for v in train_2d.step_losses(model, batch, 0, train_2d.MSELoss2d(), CosineSimilarityMean())]
    for a, b in zip(vals[torch.float32], vals[torch.bfloat16]):
        assert abs(a - b) < 3e-2 * max(1.0, abs(a)), vals


DDP2_WORKER_2D = r'''
import os, sys, random, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import pcrlv2_2d_oracle as O
from pcrlv2_amd import ddp, train_2d
from pcrlv2_amd.models import PCRLv2
from pcrlv2_amd.optim import FusedSGD
from pcrlv2_amd.train_3d import CosineSimilarityMean
rank, world, _ = ddp.init_process_group_from_env("gloo")     # two processes, ONE GPU: gloo moves the CUDA buffers
torch.cuda.set_device(0)
batches = [O.synthetic_batch(4, 64, 32, seed=50 + 10 * rank + s) for s in range(2)]     # different data per rank
finals = []
for overlap in (True, False):
    random.seed(3); torch.manual_seed(0)
    model = PCRLv2().cuda().set_compute_dtype(torch.bfloat16)
    opt = FusedSGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    dp = ddp.DataParallel(model, opt, bucket_mb=8.0, overlap=overlap)
    assert dp._active and opt.grad_scale == 0.5 and len(dp.reducer.buckets) >= 3
    for bt in batches:
        losses = train_2d.train_step(model, opt, bt, 3, train_2d.MSELoss2d(), CosineSimilarityMean())
        assert all(torch.isfinite(l) for l in losses)
    finals.append(opt.flat_p.clone())
    assert not getattr(dp, "_warned_late", False), "a gradient arrived after its bucket was reduced"
assert torch.equal(finals[0], finals[1]), (finals[0] - finals[1]).abs().max()
mine = finals[0].double().sum().reshape(1).cpu()
both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(both, mine)
assert both[0].item() == both[1].item(), both
dist.barrier()
print("OK", rank, flush=True)
dist.destroy_process_group()      # tear the group down before the interpreter exits (gloo threads alive at exit abort the process now and then)
'''


def test_data_parallel_two_ranks_one_gpu_gloo_2d(tmp_path):
    """N > 1 on the 2D model: two gloo ranks on cuda:0, different batches; bucket overlap on/off bit-identical, ranks agree."""
    import subprocess
    script = tmp_path / "ddp2d.py"
    script.write_text(DDP2_WORKER_2D)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29771", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert all("OK" in o for o in outs)


@pytest.mark.parametrize("layer", [
    # N, Hi, Wi, Ci, Co, K, stride, pad, up      (BASELINE C5: b = 64, 512 x 512, bf16)
    (64, 128, 128, 64, 64, 3, 1, 1, 0),          # layer1: brick forward / data gradient / weight gradient
    (64, 512, 512, 16, 16, 3, 1, 1, 0),          # decoder block 4 conv2: the right-sized narrow kernels
    (64, 256, 256, 32, 16, 3, 1, 1, 1),          # block 4 conv1 behind the fused upsample
    (64, 128, 128, 64, 128, 3, 2, 1, 0),         # layer2.0.conv1: stride 2, parity-class data gradient
    (64, 512, 512, 8, 64, 7, 2, 3, 0),           # stem (3 channels padded to 8)
])
def test_full_size_conv2d_adjoint_identities_bf16(layer):
    """BASELINE C5 sizes.  No CPU reference is affordable, but forward, data gradient and weight gradient -- three different kernels
    per geometry -- must be mutually adjoint:  <conv(x; w), dy> == <x, dgrad(dy; w)> == <w, wgrad(x, dy)>."""
    from pcrlv2_amd import ops2d
    N, Hi, Wi, Ci, Co, K, stride, pad, up = layer
    dt, dev = torch.bfloat16, torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops2d.new_act2(N, Hi, Wi, Ci, dt, dev)
    x.normal_(generator=g)
    Cw = 3 if K == 7 else Ci                      # the stem's weight has 3 input channels; x carries 5 zero channels
    if K == 7:
        x[:, 3:] = 0
    w = torch.randn(Co, Cw, K, K, device=dev, generator=g) * 0.05
    packed = ops2d.PackedConv2d()
    y, _, _ = ops2d.conv2d_forward(x, w, None, packed, stride, pad, up, dt)
    dy = torch.empty_like(y)
    dy.normal_(generator=g)
    dx, dw = ops2d.conv2d_backward(x, dy, w, packed, stride, pad, up, dt, need_dx=True)
    ops2d.ops.join_side_stream()      # dw is produced on the weight-gradient side stream (ops.side_wgrad): the engine joins it before the gradients are summed
    wq = w.to(dt).double()
    a = float((y.double() * dy.double()).sum())
    b_ = float((x[:, :dx.shape[1]].double() * dx.double()).sum())
    c = float((wq * dw.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    print(f"adjoint: <y,dy>={a:.6e} <x,dx>={b_:.6e} <w,dw>={c:.6e} (|y||dy|={scale:.3e})")
    assert abs(a - c) < 2e-4 * scale and abs(b_ - c) < 2e-4 * scale


def test_full_size_step_properties_2d_bf16():
    """One BASELINE-C5 step (b=64, 512x512 x2 + 6 local 96x96, bf16): finite losses in the expected ranges, finite gradients, the
    unselected deep-supervision heads receive none, and the step is deterministic (bit-identical when repeated)."""
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.models import PCRLv2
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    dev = torch.device("cuda:0")
    b = 64
    g = torch.Generator(device=dev).manual_seed(1234)
    kw = dict(generator=g, device=dev)
    x1 = torch.randn(b, 3, 512, 512, **kw)
    batch = (x1, x1 + 0.1 * torch.randn(b, 3, 512, 512, **kw), torch.rand(b, 3, 512, 512, **kw), None, [torch.randn(b, 3, 96, 96, **kw) for _ in range(6)])
    results = []
    for rep in range(2):
        torch.manual_seed(0)
        model = PCRLv2().cuda().set_compute_dtype(torch.bfloat16)
        opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        random.seed(0)
        out = train_2d.train_step(model, opt, batch, 0, train_2d.MSELoss2d(), CosineSimilarityMean())
        vals = [float(v) for v in out]
        assert all(v == v and abs(v) < 1e4 for v in vals), vals
        assert vals[1] > 0 and -1.0 <= vals[2] <= 1.0 and -1.0 <= vals[4] <= 1.0 and vals[3] > 0
        grads = [p.grad for p in model.parameters()]
        assert sum(gr is None for gr in grads) == 4 * 6          # four unselected deep-supervision heads x (conv w, b, bn w, b, conv w, b)
        assert all(torch.isfinite(gr).all() for gr in grads if gr is not None)
        results.append((vals, opt.flat_p.clone()))
        del model, opt
        torch.cuda.empty_cache()
    assert results[0][0] == results[1][0], "losses differ between identical runs (non-deterministic reduction?)"
    assert torch.equal(results[0][1], results[1][1]), "parameters differ between identical runs"


@pytest.mark.parametrize("tag", ["loss2d_b4_5scales", "loss2d_b2_nl3"])
def test_fused_cosine_terms_and_loss_tail_match_the_reference_cos_loss_2d(tag):
    """The ENGINE form of the 2D step's loss assembly -- the 1 + 2 * nlocal draws taken up front, all cosine means in one launch
    (pcrl_cosine_terms_*), the total in one (pcrl_loss_total) -- against the fixture the REFERENCE's own train_2d.cos_loss produced
    (oracle/make_golden.py --loss2d; the CPU test of the same fixture holds train_2d.assemble_losses): float32 kernels vs float64 numbers,
    1e-6 on every component; the gradient of the total w.r.t. the prediction features against float64 autograd of the restated assembly."""
    import math
    import numpy as np
    import pcrlv2_2d_oracle as O2
    from pcrlv2_amd import functions as Fn, train_2d
    from pcrlv2_amd.train_3d import _fused_cos_losses
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
    b, nl, epoch = int(fx["b"]), int(fx["nlocal"]), int(fx["epoch"])
    f1, f2, fl, mask1, masks1, gt = O2.fill_loss_inputs(b, nl, dtype=torch.float64)
    # float64 autograd of the reference-shaped assembly (the CPU test pins ITS values to the reference): gradients for the comparison below
    leaves = [t.clone().requires_grad_(True) for pair in f1 for t in pair]
    f1g = [[leaves[2 * k], leaves[2 * k + 1]] for k in range(len(f1))]
    random.seed(int(fx["seed"]))
    tot64 = train_2d.assemble_losses(f1g, f2, fl, mask1, masks1, gt, b, nl, epoch, torch.nn.MSELoss(), torch.nn.CosineSimilarity())[0]
    tot64.backward()

    dev = lambda t: t.float().cuda()
    g1 = [[dev(a).requires_grad_(True), dev(p).requires_grad_(True)] for a, p in f1]
    g2, gl = [[dev(a), dev(p)] for a, p in f2], [[dev(a), dev(p)] for a, p in fl]
    random.seed(int(fx["seed"]))
    draws = [random.randint(0, len(f1) - 1) for _ in range(1 + 2 * nl)]
    assert draws == [int(v) for v in fx["draws"]]
    cos2, k0 = _fused_cos_losses(g1, g2, gl, b, nl, draws=draws)
    assert k0 == int(fx["index2"])
    l1 = Fn.mse_loss(dev(mask1), dev(gt))
    l4raw = Fn.mse_loss(dev(masks1[k0]), dev(gt))
    beta = 0.5 * (1.0 + math.cos(math.pi * epoch / 240))
    total, l4, lg, ll = Fn.loss_tail(l1, cos2, l4raw, beta)
    for name, got in (("loss", total), ("loss1", l1), ("loss2", lg), ("loss4", l4), ("local_loss", ll)):
        assert abs(float(got) - float(fx[name])) < 1e-6, (name, float(got), float(fx[name]))
    total.backward()
    for k in range(len(f1)):
        for j in range(2):
            ref = leaves[2 * k + j].grad
            got = g1[k][j].grad
            if ref is None or float(ref.abs().max()) == 0.0:
                assert got is None or float(got.abs().max()) == 0.0, (k, j)
            else:
                err = float((got.double().cpu() - ref).abs().max()) / float(ref.abs().max())
                assert err < 1e-5, (k, j, err)


@pytest.mark.parametrize("dtype,local", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, False)])
def test_eval_mode_forward_uses_the_running_statistics_2d(dtype, local):
    """PCRLv2.eval() (what a consumer of the saved model runs, README.md:31-45; VERDICT r4: it raised NotImplementedError): two training steps move the
    running statistics away from their initial values, then `model.eval()` forward against the 2D oracle in eval mode (running statistics, nothing
    updated): float32 features / maps / reconstruction at 2e-4 relative to the largest entry, bf16 at 0.15 relative L2; the state_dict is bit-unchanged by the eval
    forward, and train() afterwards is the training path again.  (2D parity unpinned: the oracle is a restatement.)"""
    import pcrlv2_2d_oracle as O2
    from pcrlv2_amd import train_2d
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    model = _build(seed=5, dtype=dtype)
    model.train()
    opt = FusedSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    random.seed(1)
    for s in range(2):
        train_2d.train_step(model, opt, O2.synthetic_batch(4, 64, 32, seed=40 + s), 0, train_2d.MSELoss2d(), CosineSimilarityMean())
    torch.cuda.synchronize()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    assert float((before["model.encoder.bn1.running_mean"]).abs().max()) > 0          # the statistics have moved
    model.eval()
    x = O2.synthetic_batch(3, 64, 32, seed=77)[0]
    with torch.no_grad():
        outs, masks, mids = model(x.cuda(), local=local)
    torch.cuda.synchronize()
    after = model.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), k                                      # eval forward updates nothing
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in before.items()}
    with torch.no_grad():
        r_outs, r_masks, r_mids = O2.model_forward(x.double(), sd, local=local, training=False)
    tol = 2e-4 if dtype == torch.float32 else 0.15      # bf16, relative L2: measured 4e-2 on the pooled projections, 9e-2 on the reconstruction (21 bf16 layers whose normalisations use
                                                        # two-step-old running statistics, i.e. do NOT re-normalise the rounding noise the way batch statistics do)

    def close(got, ref, what):
        got = got.detach().float().cpu().double()
        if got.dim() == 4 and got.shape != ref.shape:
            got = got.permute(0, 3, 1, 2) if got.shape[-1] == ref.shape[1] else got
        if dtype == torch.float32:
            err = float((got.reshape(ref.shape) - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        else:       # bf16: relative L2 (single entries of a 20-layer bf16 chain normalised by two-step-old running statistics are off by up to 0.16 of the maximum)
            err = float((got.reshape(ref.shape) - ref).norm()) / max(float(ref.norm()), 1e-12)
        assert err < tol, (what, err)
    for i, ((pro, pre), (rpro, rpre)) in enumerate(zip(outs, r_outs)):
        close(pro, rpro, f"x_pro[{i}]")
        close(pre, rpre, f"x_pre[{i}]")
    assert (masks is None) == (r_masks is None) == local
    if not local:
        close(masks, r_masks, "reconstruction")
    assert len(mids) == 5
    for i, (m, rm) in enumerate(zip(mids, r_mids)):
        close(m, rm, f"deep-supervision map {i}")
    model.train()
    random.seed(2)
    out = train_2d.train_step(model, opt, O2.synthetic_batch(4, 64, 32, seed=50), 0, train_2d.MSELoss2d(), CosineSimilarityMean())
    assert all(math.isfinite(float(v)) for v in out)


def test_unread_deep_supervision_convolutions_write_nothing_2d():
    """functions2d.HEAD_STATS_ONLY (default on): 14 of the 15 deep-supervision heads of a step have no reader (train_2d.py:143-168: the second view's,
    the local views', and the four scales the first cos_loss did not draw) -- their 3x3 convolution runs for the statistics update of its
    BatchNorm2d only (pcrlv2_model.py:103-105), so where the kernel can leave the output out (pcrl_conv2d_fwd with y = NULL: the narrow kernel,
    <= 32 channels: the two full-resolution heads) it is not written.  The statistics come from the float32 accumulators either way: losses,
    parameters, momentum buffers and every BatchNorm buffer after three steps are BIT-identical to writing the outputs; and the stats-only
    launches are really taken (counted through the C ABI)."""
    import pcrlv2_2d_oracle as O
    from pcrlv2_amd import functions2d, train_2d
    from pcrlv2_amd._lib import lib
    from pcrlv2_amd.optim import FusedSGD
    from pcrlv2_amd.train_3d import CosineSimilarityMean
    batches = [O.synthetic_batch(4, 64, 32, seed=21 + k) for k in range(3)]
    keep, finals, nulls = functions2d.HEAD_STATS_ONLY, [], []
    L = lib()

    class Count:
        watch = {"pcrl_conv2d_fwd"}
        n = 0

        def add(self, name, args):
            self.n += args[3] is None       # y
    try:
        for on in (True, False):
            functions2d.HEAD_STATS_ONLY = on
            model = _build(seed=5, dtype=torch.bfloat16)
            opt = FusedSGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
            random.seed(2)
            c = Count()
            L.counter = c
            try:
                for bt in batches:
                    out = train_2d.train_step(model, opt, bt, 0, train_2d.MSELoss2d(), CosineSimilarityMean())
                torch.cuda.synchronize()
            finally:
                L.counter = None
            nulls.append(c.n)
            bufs = torch.cat([v.flatten().float() for k, v in sorted(model.state_dict().items()) if "running" in k or "num_batches" in k])
            finals.append(([float(o) for o in out], opt.flat_p.clone(), opt.flat_buf.clone(), bufs))
    finally:
        functions2d.HEAD_STATS_ONLY = keep
    assert nulls[0] > 0 and nulls[1] == 0, nulls
    a, b = finals
    assert a[0] == b[0], (a[0], b[0])
    for x, y, what in zip(a[1:], b[1:], ("parameters", "momentum buffers", "BatchNorm buffers")):
        assert torch.equal(x, y), what
