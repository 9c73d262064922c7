"""2D path: every HIP operator against a plain-PyTorch float64 CPU reference of the same op, through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _dev():
    return torch.device("cuda:0")


def _close(got, ref, dt, what, f32_tol=2e-5, bf_tol=2e-2):
    torch.cuda.synchronize()      # operator-level tests read results of EVERY engine stream (weight gradients are produced on the side stream, ops.side_wgrad)
    got = got.detach().double().cpu()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    denom = float(ref.norm()) + 1e-30
    err = float((got - ref).norm()) / denom
    tol = f32_tol if dt == torch.float32 else bf_tol
    assert err < tol, f"{what}: rel-L2 {err:.3e} > {tol}"


def _q(t, dt):
    """the values the kernel actually sees (bf16 rounding of the operands), as float64"""
    return t.to(dt).double()


GEOMS = [  # Ci, Co, K, stride, pad, up, H, W, bias
    (3, 64, 7, 2, 3, 0, 32, 40, False),      # ResNet stem (3 channels zero-padded to 8)
    (64, 64, 3, 1, 1, 0, 12, 20, False),     # BasicBlock conv
    (64, 128, 3, 2, 1, 0, 16, 24, False),    # stride-2 BasicBlock conv
    (64, 128, 1, 2, 0, 0, 16, 24, False),    # downsample
    (32, 16, 3, 1, 1, 1, 8, 12, False),      # decoder conv1 behind the fused nearest x2 upsample
    (16, 16, 3, 1, 1, 0, 16, 16, True),      # deep-supervision conv (bias)
    (512, 256, 3, 1, 1, 1, 2, 2, False),     # first decoder block at the bottleneck
    (16, 16, 3, 1, 1, 0, 16, 32, True),      # H % 8 == 0, W % 32 == 0: the right-sized narrow weight-gradient kernel (bf16)
    (32, 16, 3, 1, 1, 1, 8, 16, False),      # ... behind the fused upsample (block 4 conv1)
    (32, 32, 3, 1, 1, 0, 8, 64, False),      # ... 32 -> 32 (block 3)
    (16, 32, 3, 1, 1, 0, 24, 32, False),     # ... 16 -> 32
    (64, 32, 3, 1, 1, 1, 8, 16, False),      # ... 64 -> 32 behind the upsample (block 3 conv1): two 32-channel launches
    (64, 128, 3, 2, 1, 0, 15, 9, False),     # odd extents, stride 2 (one gather over all taps)
    (128, 256, 3, 2, 1, 0, 8, 12, False),    # stride 2, even extents: parity-class data gradient
    (128, 256, 1, 2, 0, 0, 8, 12, False),    # downsample, parity class (0, 0)
    (3, 64, 7, 2, 3, 0, 33, 21, False),      # odd extents, stem
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", GEOMS)
def test_conv2d_fwd_dgrad_wgrad(geom, dt):
    from pcrlv2_amd import ops2d
    Ci, Co, K, stride, pad, up, H, W, has_bias = geom
    N = 3
    g = torch.Generator().manual_seed(Ci * 1000 + Co + K)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5
    b = torch.randn(Co, generator=g) if has_bias else None
    xa = ops2d.to_act2(x.to(_dev()), dt, pad_to=8 if Ci < 8 else 0)
    wd = w.to(_dev())
    packed = ops2d.PackedConv2d()
    y, partial, rows = ops2d.conv2d_forward(xa, wd, None if b is None else b.to(_dev()), packed, stride, pad, up, dt)
    # reference on the rounded operands
    xr = _q(x, dt).requires_grad_(True)
    wr = _q(w, dt).requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    yr = F.conv2d(xin, wr, None if b is None else b.double(), stride, pad)
    _close(y, yr, dt, "conv2d fwd", bf_tol=6e-3)
    # statistics rows: sum and sum of squares per channel from the float accumulators
    st = partial.view(rows, Co, 2).double().sum(0).cpu()
    _close(st[:, 0], yr.sum((0, 2, 3)), torch.float32, "conv2d stats sum", f32_tol=2e-3 if dt == torch.bfloat16 else 1e-4)
    _close(st[:, 1], (yr * yr).sum((0, 2, 3)), torch.float32, "conv2d stats sumsq", f32_tol=2e-3 if dt == torch.bfloat16 else 1e-4)
    # backward
    dy = torch.randn(yr.shape, generator=g)
    dya = ops2d.to_act2(dy.to(_dev()), dt)
    dx, dw = ops2d.conv2d_backward(xa, dya, wd, packed, stride, pad, up, dt, need_dx=True)
    ops2d.ops.join_side_stream()      # dw is produced on the weight-gradient side stream (ops.side_wgrad): the engine joins it before the gradients are summed
    yr.backward(_q(dy, dt))
    _close(dx[:, :Ci], xr.grad, dt, "conv2d dgrad", bf_tol=6e-3)
    _close(dw, wr.grad, torch.float32, "conv2d wgrad", f32_tol=3e-5 if dt == torch.float32 else 1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("W", [24, 32])      # W % 32 == 0 (bf16): the right-sized narrow kernel, else the gather kernel
@pytest.mark.parametrize("K,Ci", [(1, 16), (3, 16)])
def test_conv2d_to3_float_output(K, Ci, W, dt):
    """deep_supervision_head.3 (1x1 -> 3) and the segmentation head (3x3 -> 3): float32 output, 3-channel gradient padded to 8"""
    from pcrlv2_amd import ops2d
    N, H, Co = 2, 16, 3
    g = torch.Generator().manual_seed(K)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5
    b = torch.randn(Co, generator=g)
    xa = ops2d.to_act2(x.to(_dev()), dt)
    packed = ops2d.PackedConv2d()
    y, _, _ = ops2d.conv2d_forward(xa, w.to(_dev()), b.to(_dev()), packed, 1, K // 2, 0, dt, want_stats=False, out_f32=True)
    assert y.dtype == torch.float32
    xr, wr = _q(x, dt).requires_grad_(True), _q(w, dt).requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), 1, K // 2)
    _close(y, yr, dt, "conv -> 3 fwd", bf_tol=1e-5)      # float32 store: only the operand rounding (already in the reference) remains
    dy = torch.randn(yr.shape, generator=g)
    dyp = ops2d.to_act2(dy.to(_dev()), dt, pad_to=8)
    dx, dw = ops2d.conv2d_backward(xa, dyp, w.to(_dev()), packed, 1, K // 2, 0, dt)
    ops2d.ops.join_side_stream()      # dw is produced on the weight-gradient side stream (ops.side_wgrad): the engine joins it before the gradients are summed
    yr.backward(_q(dy, dt))
    _close(dx, xr.grad, dt, "conv -> 3 dgrad", bf_tol=6e-3)
    _close(dw, wr.grad, torch.float32, "conv -> 3 wgrad", f32_tol=1e-4)
    db = ops2d.colsum(ops2d.to_act2(dy.to(_dev()), torch.float32, pad_to=8), N * H * W, 8, torch.float32)[:3]
    _close(db, dy.double().sum((0, 2, 3)), torch.float32, "bias grad", f32_tol=1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,W", [(16, 24), (15, 9)])
def test_maxpool2d(H, W, dt):
    from pcrlv2_amd import ops2d
    N, C = 2, 64
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, C, H, W, generator=g).to(dt)
    x[0, :, 0:3, 0:3] = 1.5            # ties: torch gives the gradient to the first maximum in scan order
    xa = ops2d.to_act2(x.to(_dev()), dt)
    y, idx = ops2d.maxpool_forward(xa, dt)
    xr = x.double().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(y.double().cpu(), yr.detach())
    dy = torch.randn(yr.shape, generator=g).to(dt)
    dx = ops2d.maxpool_backward(ops2d.to_act2(dy.to(_dev()), dt), idx, (N, H, W, C), dt)
    yr.backward(dy.double())
    _close(dx, xr.grad, dt, "maxpool2d bwd", f32_tol=1e-6, bf_tol=6e-3)


@pytest.mark.parametrize("scale", [1, 2, 4, 16])
def test_bilinear(scale):
    from pcrlv2_amd import ops2d
    N, C, H, W = 2, 3, 6, 5
    g = torch.Generator().manual_seed(scale)
    x = torch.randn(N, C, H, W, generator=g)
    xa = ops2d.to_act2(x.to(_dev()), torch.float32)
    y = ops2d.bilinear_forward(xa, scale)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=scale, mode="bilinear")
    _close(y, yr, torch.float32, "bilinear fwd", f32_tol=1e-6)
    dy = torch.randn(yr.shape, generator=g)
    dx = ops2d.bilinear_backward(ops2d.to_act2(dy.to(_dev()), torch.float32), (N, H, W, C), scale)
    yr.backward(dy.double())
    _close(dx, xr.grad, torch.float32, "bilinear bwd", f32_tol=1e-6)


@pytest.mark.parametrize("dt", DTYPES)
def test_add_relu_and_mask(dt):
    from pcrlv2_amd import ops2d
    g = torch.Generator().manual_seed(0)
    t, r = torch.randn(2, 16, 8, 8, generator=g).to(dt), torch.randn(2, 16, 8, 8, generator=g).to(dt)
    ta, ra = ops2d.to_act2(t.to(_dev()), dt), ops2d.to_act2(r.to(_dev()), dt)
    a = ops2d.add_relu_forward(ta, ra, dt)
    ref = F.relu(t.double() + r.double()).to(dt)
    assert torch.equal(a.cpu(), ref)
    da = torch.randn(2, 16, 8, 8, generator=g).to(dt)
    gk = ops2d.relu_mask_backward(ops2d.to_act2(da.to(_dev()), dt), a, dt)
    assert torch.equal(gk.cpu(), torch.where(ref > 0, da, torch.zeros_like(da)))


@pytest.mark.parametrize("Ci,Co,up,H,W,N", [(64, 64, 0, 16, 24, 4), (64, 32, 0, 8, 16, 8), (128, 64, 1, 8, 8, 4), (32, 128, 0, 24, 8, 4),
                                            # the wide-brick LDS-DMA kernel (conv_brick16.h MODE 3; W % 16 == 0, or H % 16 == 0 on its permuted axes): edge bricks in every
                                            # direction, one and several bricks, 64- and 32-channel tiles, several channel tiles, two chunks and more
                                            (64, 64, 0, 16, 32, 4), (32, 64, 0, 24, 48, 8), (128, 128, 0, 8, 16, 4), (64, 128, 0, 32, 8, 4), (128, 32, 0, 16, 16, 12),
                                            (64, 64, 1, 8, 16, 4)])     # upsampled source: stays on the 4 x 8 x 8-brick kernel
def test_conv2d_brick_path(Ci, Co, up, H, W, N):
    """3x3 / stride 1 bf16 convolutions with N % 4 == H % 8 == W % 8 == 0 and channels % 32 == 0 run on the LDS-halo brick kernels -- the wide brick
    (4 images x 8 x 16 pixels, LDS-DMA; round 5) where it tiles and the source is not upsampled, else conv_brick.hip's 4 x 8 x 8 brick -- forward
    (optionally behind the nearest x2 upsample) and data gradient; statistics rows included.  impl 0 = auto, 2 = auto without the wide brick, 1 = gather."""
    from pcrlv2_amd import ops2d
    from pcrlv2_amd._lib import lib
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(Ci + Co + up)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    xa = ops2d.to_act2(x.to(_dev()), dt)
    wd = w.to(_dev())
    xr, wr = _q(x, dt).requires_grad_(True), _q(w, dt).requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    yr = F.conv2d(xin, wr, None, 1, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(_q(dy, dt))
    outs = {}
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    wide = (not up) and (Wo % 16 == 0 or (Ho % 16 == 0 and Wo % 8 == 0))
    from pcrlv2_amd._lib import dtype_code
    assert lib().call("pcrl_conv2d_fwd_kind", N, H, W, Ci, Co, 3, 3, 1, 1, up, 0, dtype_code(dt)) == (3 if wide else 1)
    for impl in (0, 2, 1):         # 0 = auto (wide brick where it tiles), 2 = auto without the wide brick, 1 = gather kernel
        lib().debug_set_conv2d_impl(impl)
        lib().debug_set_wgrad_impl(1 if impl == 1 else 0)     # weight gradient: brick kernel (wgrad_brick.hip, nkd = 1) where Co % 64 == 0 and no upsample
        try:
            packed = ops2d.PackedConv2d()
            y, partial, rows = ops2d.conv2d_forward(xa, wd, None, packed, 1, 1, up, dt)
            dx, dw = ops2d.conv2d_backward(xa, ops2d.to_act2(dy.to(_dev()), dt), wd, packed, 1, 1, up, dt)
            ops2d.ops.join_side_stream()      # dw is produced on the weight-gradient side stream (ops.side_wgrad): the engine joins it before the gradients are summed
        finally:
            lib().debug_set_conv2d_impl(0)
            lib().debug_set_wgrad_impl(0)
        _close(dw, wr.grad, torch.float32, f"wgrad impl={impl}", f32_tol=1e-4)
        _close(y, yr, dt, f"fwd impl={impl}", bf_tol=6e-3)
        _close(dx, xr.grad, dt, f"dgrad impl={impl}", bf_tol=6e-3)
        st = partial.view(rows, Co, 2).double().sum(0).cpu()
        _close(st[:, 0], yr.sum((0, 2, 3)), torch.float32, f"stats sum impl={impl}", f32_tol=3e-3)
        _close(st[:, 1], (yr * yr).sum((0, 2, 3)), torch.float32, f"stats sumsq impl={impl}", f32_tol=3e-3)
        outs[impl] = (y.float().cpu(), dx.float().cpu())
    # same operands, same fp32 accumulation: the two kernels differ only in summation order
    assert (outs[0][0] - outs[1][0]).abs().max() <= 0.02 * outs[1][0].abs().max()
    assert (outs[0][0] - outs[2][0]).abs().max() <= 0.02 * outs[1][0].abs().max() and (outs[0][1] - outs[2][1]).abs().max() <= 0.02 * outs[1][1].abs().max()


# ---- round 4: the 3-channel ends of the step (csrc/heads2d.hip) and the summed-gradient BatchNorm backward ----------------------------
@pytest.mark.parametrize("N,H,W", [(2, 16, 24), (3, 37, 29), (1, 64, 64)])
def test_mse2d_against_torch(N, H, W):
    """pcrl_mse2d_fwd / _bwd_pad: NHWC float32 prediction against the NCHW image -- loss, the padded gradient (bf16 / f32, CP = 8; f32 CP = 3)
    and its per-channel sums."""
    from pcrlv2_amd import ops2d
    g = torch.Generator().manual_seed(N * H + W)
    p = torch.randn(N, 3, H, W, generator=g, dtype=torch.float64)
    gt = torch.rand(N, 3, H, W, generator=g, dtype=torch.float64)
    pa = ops2d.to_act2(p.float().to(_dev()), torch.float32)
    gd = gt.float().to(_dev())
    loss = ops2d.mse2d_forward(pa, gd)
    pr = p.float().double().requires_grad_(True)
    ref = F.mse_loss(pr, gt.float().double())
    assert abs(float(loss) - float(ref.detach())) < 1e-6 * max(1.0, float(ref.detach()))
    dl = torch.tensor(0.7, device=_dev())
    (ref * 0.7).backward()
    for dt, CP, tol in ((torch.float32, 3, 2e-6), (torch.float32, 8, 2e-6), (torch.bfloat16, 8, 5e-3)):
        dy, colpart, rows = ops2d.mse2d_backward(pa, gd, dl, CP, dt)
        assert tuple(dy.shape) == (N, CP, H, W) and dy.dtype == dt
        _close(dy[:, :3], pr.grad, torch.float32, f"mse2d bwd {dt} CP={CP}", f32_tol=tol)
        if CP > 3:
            assert float(dy[:, 3:].abs().max()) == 0.0
            sums = ops2d.colsum_f32(colpart, rows, CP)
            assert float(sums[3:].abs().max()) == 0.0
            _close(sums[:3], dy[:, :3].double().sum((0, 2, 3)).cpu(), torch.float32, "bias sums = sums of the stored (rounded) gradient", f32_tol=2e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Ci,N,H,W", [(16, 2, 32, 32), (32, 2, 24, 40), (64, 3, 16, 16), (256, 2, 8, 8), (128, 1, 5, 7)])
def test_conv1x1_to3_backward_one_pass(Ci, N, H, W, dt):
    """pcrl_conv2d_1x1_small_bwd: dx, dw, db of nn.Conv2d(Ci, 3, 1) from one pass, against float64 autograd."""
    from pcrlv2_amd import ops2d
    g = torch.Generator().manual_seed(Ci + H)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(3, Ci, 1, 1, generator=g) / Ci ** 0.5
    dy = torch.randn(N, 3, H, W, generator=g)
    xa = ops2d.to_act2(x.to(_dev()), dt)
    dya = ops2d.to_act2(dy.to(_dev()), torch.float32)
    dx, dw, db = ops2d.conv1x1_small_backward(xa, dya, w.to(_dev()), dt)
    xr, wr = _q(x, dt).requires_grad_(True), w.double().requires_grad_(True)
    br = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, br).backward(dy.double())
    _close(dx, xr.grad, dt, "1x1 -> 3 dgrad", bf_tol=6e-3)
    _close(dw, wr.grad, torch.float32, "1x1 -> 3 wgrad", f32_tol=2e-5)
    _close(db, br.grad, torch.float32, "1x1 -> 3 bias grad", f32_tol=2e-5)


@pytest.mark.parametrize("dt", DTYPES)
def test_image_to_act_pads_and_transposes(dt):
    from pcrlv2_amd import ops2d
    x = torch.randn(3, 3, 20, 28)
    got = ops2d.image_to_act(x.to(_dev()), dt, 8)
    assert tuple(got.shape) == (3, 8, 20, 28) and got.dtype == dt
    ops2d.dims2(got)     # NHWC memory
    assert torch.equal(got[:, :3].float().cpu(), x.to(dt).float())
    assert float(got[:, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("parts", ["da+da2", "da+da2+row", "da2+row", "da+row"])
@pytest.mark.parametrize("C,N,S", [(16, 4, 96), (64, 3, 40), (256, 2, 64)])
def test_bn_backward_of_a_summed_gradient(C, N, S, parts, dt):
    """pcrl_bn_act_bwd_{reduce,apply}_sum: the BatchNorm backward fed with da + da2 + row_g / S in parts equals the backward fed with the
    materialised sum (same kernels, one tensor): dy to one ulp of the storage type, dgamma / dbeta to float32 round-off."""
    from pcrlv2_amd import ops
    from pcrlv2_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(C + S)
    M = N * S
    dev = _dev()
    y = torch.randn(M, C, generator=g).to(dev).to(dt)
    da = torch.randn(M, C, generator=g).to(dev).to(dt)
    da2 = torch.randn(M, C, generator=g).to(dev).to(dt)
    row = torch.randn(N, C, generator=g).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    mean = y.float().mean(0)
    var = y.float().var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    scale = gamma * rstd
    shift = 0.1 - mean * scale
    use_da, use_da2, use_row = "da+" in parts or parts.startswith("da+"), "da2" in parts, "row" in parts
    use_da = parts.split("+")[0] == "da"
    full = torch.zeros(M, C, device=dev)
    if use_da:
        full += da.float()
    if use_da2:
        full += da2.float()
    if use_row:
        full += (row / S).repeat_interleave(S, dim=0)
    args = (gamma, mean, rstd, scale, shift, M, C, ACT_RELU)
    # reference: float64 BatchNorm backward of the exact sum (what the kernels accumulate in float32 from the parts)
    yd, fd = y.double(), full.double()
    z = scale.double() * yd + shift.double()
    dz = torch.where(z > 0, fd, torch.zeros_like(fd))
    xhat = (yd - mean.double()) * rstd.double()
    dbeta_ref, dgamma_ref = dz.sum(0), (dz * xhat).sum(0)
    dy_ref = gamma.double() * rstd.double() * (dz - dbeta_ref / M - xhat * dgamma_ref / M)
    a = da if use_da else None
    b = da2 if use_da2 else None
    if a is None:
        a, b = b, None
    dy, dgamma, dbeta = ops.bn_act_backward(a, y, *args, dt, row_g=row if use_row else None, da2=b)
    _close(dy, dy_ref.cpu(), dt, "dy", f32_tol=2e-5, bf_tol=6e-3)
    _close(dgamma, dgamma_ref.cpu(), torch.float32, "dgamma", f32_tol=1e-4)
    _close(dbeta, dbeta_ref.cpu(), torch.float32, "dbeta", f32_tol=1e-4)


@pytest.mark.parametrize("N,H,W", [(2, 32, 128), (3, 48, 64), (1, 16, 192)])
def test_stem_kernels_against_torch(N, H, W):
    """csrc/stem2d.hip: Conv2d(3, 64, 7, stride 2, padding 3) on the float32 NCHW image -- forward, BatchNorm statistics rows and the weight
    gradient against float64 torch on the bf16-rounded operands; and against the general gather kernel on the padded image (same operands)."""
    from pcrlv2_amd import ops2d
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    xd, wd = x.to(_dev()), w.to(_dev())
    assert ops2d.stem_ok(xd, wd, dt)
    y, partial, rows = ops2d.stem_forward(xd, wd, ops2d.PackedStem(), dt)
    xr, wr = _q(x, dt).requires_grad_(True), _q(w, dt).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, 3)
    _close(y, yr, dt, "stem fwd", bf_tol=4e-3)
    st = partial.view(rows, 64, 2).double().sum(0).cpu()
    _close(st[:, 0], yr.detach().sum((0, 2, 3)), torch.float32, "stem statistics: sums", f32_tol=1e-3)
    _close(st[:, 1], (yr.detach() ** 2).sum((0, 2, 3)), torch.float32, "stem statistics: sums of squares", f32_tol=1e-4)
    dy = torch.randn(yr.shape, generator=g)
    dya = ops2d.to_act2(dy.to(_dev()), dt)
    dw = ops2d.stem_wgrad(xd, dya, wd, dt)
    torch.cuda.synchronize()
    yr.backward(_q(dy, dt))
    _close(dw, wr.grad, torch.float32, "stem wgrad", f32_tol=2e-4)
    # the gather kernel on the image padded to 8 channels computes the same thing from the same rounded operands
    packed = ops2d.PackedConv2d()
    y2, _, _ = ops2d.conv2d_forward(ops2d.image_to_act(xd, dt, 8), wd, None, packed, 2, 3, 0, dt)
    _close(y, y2.double().cpu(), dt, "stem fwd vs gather kernel", bf_tol=4e-3)


@pytest.mark.parametrize("N,C", [(4, 16), (32, 128), (64, 256), (192, 64), (384, 32), (500, 16)])
@pytest.mark.parametrize("parts", ["pro+pre", "pre", "pro"])
def test_heads_backward_two_launches_equal_the_nine(N, C, parts):
    """ops.heads_backward through pcrl_head_bwd_stage (Linear backward + the BatchNorm1d backward of what it produced, one launch per half of
    the chain) against the separate kernels (ops.FUSED_HEAD_BWD = False) and against float64 autograd of the same chain."""
    from pcrlv2_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(N + C)
    H = 2 * C
    pooled = torch.randn(N, C, generator=g)
    P = dict(bn_g=torch.rand(C, generator=g) + 0.5, bn_b=torch.randn(C, generator=g) * 0.1, p0_w=torch.randn(H, C, generator=g) / C ** 0.5,
             p0_b=torch.randn(H, generator=g) * 0.1, p1_g=torch.rand(H, generator=g) + 0.5, p1_b=torch.randn(H, generator=g) * 0.1,
             p3_w=torch.randn(C, H, generator=g) / H ** 0.5, p3_b=torch.randn(C, generator=g) * 0.1)
    d_pro = torch.randn(N, C, generator=g) if "pro" in parts else None
    d_pre = torch.randn(N, C, generator=g) if "pre" in parts else None
    # float64 reference
    R = {k: v.double().requires_grad_(True) for k, v in P.items()}
    gp = pooled.double().requires_grad_(True)
    x_pro = F.batch_norm(gp, None, None, R["bn_g"], R["bn_b"], True, 0.1, 1e-5)
    h0 = x_pro @ R["p0_w"].t() + R["p0_b"]
    h1 = torch.relu(F.batch_norm(h0, None, None, R["p1_g"], R["p1_b"], True, 0.1, 1e-5))
    x_pre = h1 @ R["p3_w"].t() + R["p3_b"]
    loss = 0.0
    if d_pro is not None:
        loss = loss + (x_pro * d_pro.double()).sum()
    if d_pre is not None:
        loss = loss + (x_pre * d_pre.double()).sum()
    loss.backward()
    # engine forward pieces (the saved tensors of the stage Functions)
    D = {k: v.to(dev) for k, v in P.items()}
    pg = pooled.to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    xp, m_pro, r_pro = ops.bn1d_forward(pg, D["bn_g"], D["bn_b"], rm, rv, relu=False)
    e_h0 = ops.linear_forward(xp, D["p0_w"], D["p0_b"])
    e_h1, m_h, r_h = ops.bn1d_forward(e_h0, D["p1_g"], D["p1_b"], torch.zeros(H, device=dev), torch.ones(H, device=dev), relu=True)
    heads = (pg, m_pro, r_pro, e_h0, e_h1, m_h, r_h)
    params = tuple(D[k] for k in ("bn_g", "bn_b", "p0_w", "p0_b", "p1_g", "p1_b", "p3_w", "p3_b"))
    dp = d_pro.to(dev) if d_pro is not None else None
    dq = d_pre.to(dev) if d_pre is not None else None
    outs = []
    keep = ops.FUSED_HEAD_BWD
    try:
        for fused in (True, False):
            ops.FUSED_HEAD_BWD = fused
            outs.append(ops.heads_backward(dp, dq, heads, xp, params))
    finally:
        ops.FUSED_HEAD_BWD = keep
    names = ("bn_g", "bn_b", "p0_w", "p0_b", "p1_g", "p1_b", "p3_w", "p3_b")
    for d_g, grads in outs:
        _close(d_g, gp.grad, torch.float32, "d pooled", f32_tol=2e-4)
        for nm, gr in zip(names, grads):
            if R[nm].grad is None or (d_pre is None and nm.startswith(("p0", "p1", "p3"))):
                assert gr is None, nm
                continue
            ref = R[nm].grad
            if float(ref.norm()) < 1e-9:        # a bias in front of a BatchNorm1d: identically zero
                assert float(gr.abs().max()) < 1e-4, nm
            else:
                _close(gr, ref, torch.float32, nm, f32_tol=2e-4)
