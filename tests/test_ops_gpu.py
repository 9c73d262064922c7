"""GPU parity of every kernel family, called through the C ABI (pcrlv2_amd.ops -> libpcrl_hip.so), against a plain
PyTorch float64 CPU computation of the same operator.

Tolerances: float32 mode -- 2e-5 * max|ref| (accumulation-order noise of an exact-fp32 MFMA chain);
bfloat16 mode -- operands are pre-rounded to bf16 on both sides, so what remains is fp32 accumulation order plus ONE
bf16 rounding of the stored result: 1e-2 * max|ref| (bf16 has 8 mantissa bits: 2^-8 = 3.9e-3 relative).
Reductions that stay in float32 on both sides (weight gradients, statistics) use the tight bound in both modes.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, CONV_BM, PcrlError, dtype_code, lib, stream_handle  # noqa: E402

DEV = "cuda"
DEV_T = torch.device("cuda")
DTYPES = [torch.float32, torch.bfloat16]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def q(t, dt):
    """round a float64 CPU tensor to the activation dtype and back (what the kernel sees)"""
    return t.to(dt).double()


def act_dev(t, dt):
    """float64 NCDHW CPU tensor -> NDHWC device activation in dt"""
    return ops.to_act(t.to(dt).to(DEV), dt)


def back(t):
    torch.cuda.synchronize()      # operator-level tests read results of EVERY engine stream (weight gradients are produced on the side stream, ops.side_wgrad)
    return t.detach().double().cpu().contiguous()


def check(got, ref, dt, what, out_rounded=True, f32_tol=2e-5, bf_tol=1e-2):
    got, ref = back(got) if torch.is_tensor(got) and got.is_cuda else got.double(), ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-6)
    tol = (bf_tol if (dt == torch.bfloat16 and out_rounded) else f32_tol) * scale
    err = (got - ref).abs().max().item()
    assert err <= tol, f"{what} [{dt}]: max|d|={err:.3e} > tol {tol:.3e} (ref max {scale:.3e})"
    return err


SHAPES3 = [  # N, D, H, W, Ci, Co   (M deliberately not a multiple of 128 in the first rows)
    (3, 6, 5, 7, 32, 64), (2, 4, 4, 4, 64, 32), (1, 8, 8, 4, 64, 128), (2, 2, 2, 2, 128, 256), (1, 16, 8, 8, 32, 64),
    # shapes served by the LDS-halo brick kernel in bf16 (D%4 == H%8 == W%8 == 0): edge bricks in every direction
    (2, 8, 16, 8, 64, 64), (1, 4, 8, 16, 96, 128), (3, 4, 8, 8, 32, 64), (1, 12, 24, 16, 64, 64), (2, 2, 8, 8, 128, 64),
    # several channel tiles: the XCD co-located launches (output-channel tiles of a brick in conv_brick.hip; kd planes x tile pairs of a
    # brick range in wgrad_brick.hip: pairing over ci tiles, over co tiles, both possible) and more than one brick range
    (2, 8, 16, 16, 64, 128), (2, 8, 16, 16, 128, 64), (1, 8, 16, 16, 128, 128), (1, 8, 16, 16, 64, 192), (3, 16, 16, 16, 32, 64),
    (1, 4, 16, 16, 256, 128),
    # 4x8x16-brick kernel (conv_brick16.hip): edge bricks in d, h, w; one and several bricks per direction; BN = 32 (Co = 32), Co = 96
    (1, 8, 16, 32, 32, 64), (2, 4, 8, 16, 64, 32), (1, 12, 24, 48, 32, 96), (2, 8, 8, 32, 96, 64),
    # the same kernel with its brick axes along (D, W, H): H % 16 == 0, W % 8 == 0 but not W % 16 (the 16 x 16 x 8 level)
    (3, 4, 16, 8, 64, 64), (1, 4, 32, 8, 64, 128), (2, 8, 16, 24, 32, 96), (1, 16, 16, 8, 128, 32),
    # brick kernels with the innermost extent as the 4-deep brick axis (the 8 x 8 x 4 bottleneck level; W = 12: three bricks along W)
    (2, 8, 8, 4, 64, 128), (3, 16, 8, 4, 32, 64), (1, 8, 16, 12, 64, 64),
    # gather kernel with voxel-major rows and per-tile tap skipping (volumes of <= 8 voxels, M >= 128): tiles of one, two and many voxels
    (40, 2, 2, 2, 64, 64), (48, 4, 4, 4, 32, 64), (200, 2, 2, 2, 32, 32), (33, 2, 2, 1, 64, 32), (130, 1, 1, 1, 32, 32), (70, 1, 2, 2, 32, 64),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", SHAPES3)
def test_conv3d_fwd_stats_dgrad_wgrad(shape, dt):
    N, D, H, W, Ci, Co = shape
    L, s = lib(), stream_handle()
    x, w, b = rnd(N, Ci, D, H, W, seed=1), rnd(Co, Ci, 3, 3, 3, seed=2, scale=0.1), rnd(Co, seed=3)
    xq, wq = q(x, dt), q(w, dt)
    ref = F.conv3d(xq, wq, b, padding=1)
    pk = ops.PackedWeights("conv3")
    wdev = w.float().to(DEV)
    wf, wd = pk.get(wdev, dt)
    xa = act_dev(x, dt)
    M = N * D * H * W
    rows = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
    y = ops.new_act(N, D, H, W, Co, dt, DEV)
    part = torch.zeros(rows * Co * 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y, part, N, D, H, W, Ci, Co, dtype_code(dt), s)
    check(y, ref, dt, "conv3 fwd")
    st = back(part).view(rows, Co, 2).sum(0)
    check(st[:, 0], ref.sum(dim=(0, 2, 3, 4)), dt, "conv3 stats sum", out_rounded=False, f32_tol=1e-4)
    check(st[:, 1], (ref * ref).sum(dim=(0, 2, 3, 4)), dt, "conv3 stats sumsq", out_rounded=False, f32_tol=1e-4)
    # data gradient == conv with flipped / transposed weights
    dy = rnd(N, Co, D, H, W, seed=4)
    dyq = q(dy, dt)
    xr = xq.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    F.conv3d(xr, wr, None, padding=1).backward(dyq)
    dya = act_dev(dy, dt)
    dx = ops.new_act(N, D, H, W, Ci, dt, DEV)
    L.call("pcrl_conv3d_k3_fwd", dya, wd, None, dx, None, N, D, H, W, Co, Ci, dtype_code(dt), s)
    check(dx, xr.grad, dt, "conv3 dgrad")
    # weight gradient (float32 out in both modes): both bf16 fragment-fetch paths
    nb = L.call("pcrl_conv3d_k3_wgrad_ws_bytes", N, D, H, W, Ci, Co)
    # impl 0 = auto (LDS-halo brick kernel where eligible, XCD co-located launch, tile shape by channel counts), 6 = the same with
    # 64 x 64 tiles only, 2 = the brick kernel on its plain 2-D grid, 1 = gather kernel; tr = bf16 fragment fetch of the gather kernel
    for impl, tr in (((0, 1), (6, 1), (2, 1), (1, 1), (1, 0)) if dt == torch.bfloat16 else ((0, 1),)):
        L.debug_set_wgrad_impl(impl)
        L.debug_set_wgrad_tr(tr)
        dw = torch.zeros(Co, Ci, 3, 3, 3, dtype=torch.float32, device=DEV)
        L.call("pcrl_conv3d_k3_wgrad", xa, dya, dw, ops.workspace(nb, xa.device), nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
        check(dw, wr.grad, dt, f"conv3 wgrad impl={impl} tr={tr}", out_rounded=False, f32_tol=3e-5)
    L.debug_set_wgrad_tr(1)
    L.debug_set_wgrad_impl(0)
    # the gather forward kernel on the same shape (impl 1), in case impl 0 took the brick kernel above
    L.debug_set_conv_impl(1)
    rows1 = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
    part1 = torch.zeros(rows1 * Co * 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y, part1, N, D, H, W, Ci, Co, dtype_code(dt), s)
    L.debug_set_conv_impl(0)
    check(y, ref, dt, "conv3 fwd (gather kernel)")
    check(back(part1).view(rows1, Co, 2).sum(0)[:, 0], ref.sum(dim=(0, 2, 3, 4)), dt, "conv3 stats (gather)", out_rounded=False, f32_tol=1e-4)
    if dt == torch.bfloat16 and D % 4 == 0 and H % 8 == 0 and W % 16 == 0:   # the 4x8x8-brick kernel where impl 0 took the 4x8x16 one
        L.debug_set_conv_impl(4)
        rows4 = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
        part4 = torch.zeros(rows4 * Co * 2, dtype=torch.float32, device=DEV)
        y4 = ops.new_act(N, D, H, W, Co, dt, DEV)
        L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y4, part4, N, D, H, W, Ci, Co, dtype_code(dt), s)
        dx4 = ops.new_act(N, D, H, W, Ci, dt, DEV)
        L.call("pcrl_conv3d_k3_fwd", dya, wd, None, dx4, None, N, D, H, W, Co, Ci, dtype_code(dt), s)
        L.debug_set_conv_impl(0)
        assert rows4 == 2 * rows
        check(y4, ref, dt, "conv3 fwd (4x8x8 bricks)")
        check(dx4, xr.grad, dt, "conv3 dgrad (4x8x8 bricks)")
        check(back(part4).view(rows4, Co, 2).sum(0)[:, 1], (ref * ref).sum(dim=(0, 2, 3, 4)), dt, "conv3 stats (4x8x8 bricks)", out_rounded=False, f32_tol=1e-4)
    if dt == torch.bfloat16 and D % 8 == 0 and Co % 64 == 0 and (H % 8 == 0 and W % 16 == 0 or H % 16 == 0 and W % 8 == 0):
        # wide-brick kernel on 8 x 8 x 16 bricks (one eight-wave block; impl 6) vs 4 x 8 x 16 bricks (impl 5): same accumulation order per voxel and
        # the same statistics rows (one per 4-plane half) -> bit-identical output, data gradient and partial statistics
        outs = []
        for impl in (5, 6):
            L.debug_set_conv_impl(impl)
            rows5 = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
            part5 = torch.zeros(rows5 * Co * 2, dtype=torch.float32, device=DEV)
            y5 = ops.new_act(N, D, H, W, Co, dt, DEV)
            L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y5, part5, N, D, H, W, Ci, Co, dtype_code(dt), s)
            dx5 = ops.new_act(N, D, H, W, Ci, dt, DEV)
            if Ci % 64 == 0:
                L.call("pcrl_conv3d_k3_fwd", dya, wd, None, dx5, None, N, D, H, W, Co, Ci, dtype_code(dt), s)
            else:
                dx5.zero_()
            outs.append((y5, part5, dx5))
        L.debug_set_conv_impl(0)
        check(outs[1][0], ref, dt, "conv3 fwd (8x8x16 bricks)")
        assert all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1])), "8-plane bricks differ from 4-plane bricks"
    if dt == torch.bfloat16 and Co > 64:   # the brick kernel on its 2-D grid (impl 3): bit-identical to the co-located launch
        y3 = ops.new_act(N, D, H, W, Co, dt, DEV)
        part3 = torch.zeros(rows * Co * 2, dtype=torch.float32, device=DEV)
        L.debug_set_conv_impl(4)    # both launches on the 4x8x8-brick kernel: 2-D grid vs co-located
        rows3 = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
        part3 = torch.zeros(rows3 * Co * 2, dtype=torch.float32, device=DEV)
        part3b = torch.zeros_like(part3)
        L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y, part3b, N, D, H, W, Ci, Co, dtype_code(dt), s)
        L.debug_set_conv_impl(3)
        L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y3, part3, N, D, H, W, Ci, Co, dtype_code(dt), s)
        L.debug_set_conv_impl(0)
        assert torch.equal(y3, y) and torch.equal(part3, part3b)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(4, 8, 4, 4, 64, 128), (3, 2, 2, 2, 96, 64), (2, 4, 4, 4, 256, 32),   # (8x8x4 runs on the brick kernel: axis permutation)
                                   (40, 2, 2, 2, 128, 64), (24, 4, 4, 4, 128, 128), (192, 2, 2, 2, 256, 128)])  # voxel-major rows: the splits cut each tile's own tap list
def test_conv3d_small_volume_split_k(shape, dt):
    """pcrl_conv3d_k3_fwd_ws: the K-split gather path of small volumes (8x8x4 bottleneck, 4^3 / 2^3 local-view levels) must give
    the one-pass result: output, bias, and the BatchNorm partial statistics (same row count)."""
    N, D, H, W, Ci, Co = shape
    L, s = lib(), stream_handle()
    x, w, b = rnd(N, Ci, D, H, W, seed=11), rnd(Co, Ci, 3, 3, 3, seed=12, scale=0.1), rnd(Co, seed=13)
    ref = F.conv3d(q(x, dt), q(w, dt), b, padding=1)
    wf, _ = ops.PackedWeights("conv3").get(w.float().to(DEV), dt)
    xa = act_dev(x, dt)
    rows = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
    nb = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dt))
    assert nb > 0, "these shapes are expected to take the split-K path"
    y = ops.new_act(N, D, H, W, Co, dt, DEV)
    part = torch.zeros(rows * Co * 2, dtype=torch.float32, device=DEV)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    L.call("pcrl_conv3d_k3_fwd_ws", xa, wf, b.float().to(DEV), y, part, ws, nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
    check(y, ref, dt, "conv3 fwd (split-K)")
    st = back(part).view(rows, Co, 2).sum(0)
    check(st[:, 0], ref.sum(dim=(0, 2, 3, 4)), dt, "conv3 stats sum (split-K)", out_rounded=False, f32_tol=1e-4)
    check(st[:, 1], (ref * ref).sum(dim=(0, 2, 3, 4)), dt, "conv3 stats sumsq (split-K)", out_rounded=False, f32_tol=1e-4)
    # one-pass kernel on the same input: same rounding points except the K-summation order
    y1 = ops.new_act(N, D, H, W, Co, dt, DEV)
    L.call("pcrl_conv3d_k3_fwd", xa, wf, b.float().to(DEV), y1, None, N, D, H, W, Ci, Co, dtype_code(dt), s)
    d = (back(y).float() - back(y1).float()).abs().max().item()
    assert d <= (2e-4 if dt == torch.float32 else 0.05), d
    with pytest.raises(PcrlError):
        L.call("pcrl_conv3d_k3_fwd_ws", xa, wf, None, y, None, ws, 16, N, D, H, W, Ci, Co, dtype_code(dt), s)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 4, 5, 64, 64), (1, 4, 4, 2, 128, 128), (3, 2, 2, 2, 32, 32), (2, 5, 4, 4, 256, 128), (1, 3, 4, 4, 512, 64)])
def test_convt_k2s2(shape, dt):
    N, D, H, W, Ci, Co = shape
    x, w, b = rnd(N, Ci, D, H, W, seed=1), rnd(Ci, Co, 2, 2, 2, seed=2, scale=0.2), rnd(Co, seed=3)
    xq, wq = q(x, dt).requires_grad_(True), q(w, dt).requires_grad_(True)
    bq = b.clone().requires_grad_(True)
    ref = F.conv_transpose3d(xq, wq, bq, stride=2)
    dy = rnd(N, Co, 2 * D, 2 * H, 2 * W, seed=5)
    ref.backward(q(dy, dt))
    pk = ops.PackedWeights("convt")
    wdev, bdev = w.float().to(DEV), b.float().to(DEV)
    xa = act_dev(x, dt)
    y = ops.convt_forward(xa, wdev, bdev, pk, dt)
    check(y, ref, dt, "convT fwd")
    dx, dw, db = ops.convt_backward(xa, act_dev(dy, dt), wdev, pk, dt)
    check(dx, xq.grad, dt, "convT dgrad")
    check(dw, wq.grad, dt, "convT wgrad", out_rounded=False, f32_tol=3e-5)
    check(db, bq.grad, dt, "convT bias grad", out_rounded=False, f32_tol=3e-5)


@pytest.mark.parametrize("dt", DTYPES)
def test_first_layer_c1(dt):
    N, D, H, W, Co = 3, 6, 5, 7, 32
    L, s = lib(), stream_handle()
    x, w, b = rnd(N, 1, D, H, W, seed=1), rnd(Co, 1, 3, 3, 3, seed=2), rnd(Co, seed=3)
    wr = w.clone().requires_grad_(True)
    ref = F.conv3d(x, wr, b, padding=1)
    M = N * D * H * W
    rows = L.call("pcrl_conv3d_k3_c1_stats_rows", N, D, H, W, Co, dtype_code(dt))
    assert rows == (M + CONV_BM - 1) // CONV_BM
    y = ops.new_act(N, D, H, W, Co, dt, DEV)
    part = torch.zeros(rows * Co * 2, dtype=torch.float32, device=DEV)
    xd = x.float().to(DEV)
    L.call("pcrl_conv3d_k3_c1_fwd", xd, w.float().to(DEV), b.float().to(DEV), y, part, N, D, H, W, Co, dtype_code(dt), s)
    check(y, ref, dt, "c1 fwd")
    st = back(part).view(rows, Co, 2).sum(0)
    check(st[:, 0], ref.detach().sum(dim=(0, 2, 3, 4)), dt, "c1 stats", out_rounded=False, f32_tol=1e-4)
    dy = rnd(N, Co, D, H, W, seed=4)
    # the weight gradient is an MFMA GEMM whose operands (dy and the im2col of x) are stored in the activation dtype
    F.conv3d(q(x, dt), wr, b, padding=1).backward(q(dy, dt))
    nb = L.call("pcrl_conv3d_k3_c1_wgrad_ws_bytes", N, D, H, W, Co)
    dw = torch.zeros(Co, 1, 3, 3, 3, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_k3_c1_wgrad", xd, act_dev(dy, dt), dw, ops.workspace(nb, xd.device), nb, N, D, H, W, Co, dtype_code(dt), s)
    check(dw, wr.grad, dt, "c1 wgrad", out_rounded=False, f32_tol=3e-5)


@pytest.mark.parametrize("Co", [32, 64])
def test_first_layer_c1_brick_kernel(Co):
    """bf16 first-layer forward on a brick-eligible volume: the MFMA kernel (27 taps = one K step, im2col built from the LDS
    halo) against F.conv3d with x and w rounded to bf16, and its per-brick BatchNorm partials."""
    dt = torch.bfloat16
    N, D, H, W = 2, 8, 16, 24
    L, s = lib(), stream_handle()
    x, w, b = rnd(N, 1, D, H, W, seed=41), rnd(Co, 1, 3, 3, 3, seed=42), rnd(Co, seed=43)
    ref = F.conv3d(q(x, dt), q(w, dt), b, padding=1)
    rows = L.call("pcrl_conv3d_k3_c1_stats_rows", N, D, H, W, Co, dtype_code(dt))
    assert rows == N * (D // 4) * (H // 8) * (W // 8)
    y = ops.new_act(N, D, H, W, Co, dt, DEV)
    part = torch.zeros(rows * Co * 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_k3_c1_fwd", x.float().to(DEV), w.float().to(DEV), b.float().to(DEV), y, part, N, D, H, W, Co, dtype_code(dt), s)
    check(y, ref, dt, "c1 fwd (brick)")
    st = back(part).view(rows, Co, 2).sum(0)
    check(st[:, 0], ref.sum(dim=(0, 2, 3, 4)), dt, "c1 stats sum (brick)", out_rounded=False)
    check(st[:, 1], (ref * ref).sum(dim=(0, 2, 3, 4)), dt, "c1 stats sumsq (brick)", out_rounded=False)


@pytest.mark.parametrize("case", ["c1:32", "to1:64", "to1:128", "to1:256"])
def test_one_channel_wgrad_brick_kernel(case):
    """bf16 weight gradients of the 1-channel convolutions on a brick-eligible volume (in-LDS im2col of the scalar operand):
    first layer dw[c][t] = sum dy[m][c] x[m+delta_t]; heads dw[c][t] = sum x[m][c] dy[m-delta_t] (+ bias gradient); against
    autograd with the operands rounded as the kernel sees them, and against the im2col-in-HBM path (debug switch)."""
    dt = torch.bfloat16
    kind, C = case.split(":")
    C = int(C)
    N, D, H, W = 2, 8, 16, 24
    L, s = lib(), stream_handle()
    if kind == "c1":
        x, w = rnd(N, 1, D, H, W, seed=31), rnd(C, 1, 3, 3, 3, seed=32)
        dy = rnd(N, C, D, H, W, seed=33)
        wr = w.clone().requires_grad_(True)
        F.conv3d(q(x, dt), wr, None, padding=1).backward(q(dy, dt))
        xd, dyd = x.float().to(DEV), act_dev(dy, dt)
        nb = L.call("pcrl_conv3d_k3_c1_wgrad_ws_bytes", N, D, H, W, C)
        run = lambda out: L.call("pcrl_conv3d_k3_c1_wgrad", xd, dyd, out, ops.workspace(nb, DEV_T), nb, N, D, H, W, C, dtype_code(dt), s)
        shape = (C, 1, 3, 3, 3)
    else:
        x, w = rnd(N, C, D, H, W, seed=34), rnd(1, C, 3, 3, 3, seed=35, scale=0.2)
        dy = rnd(N, 1, D, H, W, seed=36)
        wr = w.clone().requires_grad_(True)
        F.conv3d(q(x, dt), wr, None, padding=1).backward(q(dy, dt))   # dy enters the MFMA as bf16 too
        xa, dyd = act_dev(x, dt), dy.float().reshape(-1).to(DEV)
        nb = L.call("pcrl_conv3d_to1_wgrad_ws_bytes", N, D, H, W, C, 27)
        db = torch.zeros(1, dtype=torch.float32, device=DEV)
        run = lambda out: L.call("pcrl_conv3d_to1_wgrad", xa, dyd, out, db, ops.workspace(nb, DEV_T), nb, N, D, H, W, C, 27, dtype_code(dt), s)
        shape = (1, C, 3, 3, 3)
    dw = torch.zeros(shape, dtype=torch.float32, device=DEV)
    run(dw)
    check(dw, wr.grad, dt, f"{kind} wgrad (brick)", out_rounded=False, f32_tol=3e-5)
    if kind == "to1":
        assert abs(db.item() - dy.sum().item()) <= 1e-4 * max(1.0, dy.abs().sum().item())
    dw2 = torch.zeros(shape, dtype=torch.float32, device=DEV)
    L.debug_set_wgrad_impl(1)
    try:
        run(dw2)
    finally:
        L.debug_set_wgrad_impl(0)
    assert (dw - dw2).abs().max().item() <= 2e-3 * max(1.0, dw2.abs().max().item())


@pytest.mark.parametrize("C", [32, 64, 96, 128, 256])
def test_conv_to_one_channel_brick_kernel(C):
    """bf16 deep-supervision head forward on a brick-eligible volume (several bricks per axis: interior and boundary halos):
    the LDS-halo kernel against F.conv3d with bf16-rounded operands, the per-brick BatchNorm partials, and the two-pass kernel."""
    dt = torch.bfloat16
    N, D, H, W = 2, 8, 16, 24
    L, s = lib(), stream_handle()
    x, w, b = rnd(N, C, D, H, W, seed=21), rnd(1, C, 3, 3, 3, seed=22, scale=0.2), rnd(1, seed=23)
    ref = F.conv3d(q(x, dt), q(w, dt), b, padding=1)
    M = N * D * H * W
    xa = act_dev(x, dt)
    wdev, bdev = w.float().to(DEV), b.float().to(DEV)
    rows = L.call("pcrl_conv3d_to1_stats_rows", N, D, H, W, C, 27, dtype_code(dt))
    assert rows == N * (D // 4) * (H // 8) * (W // 8)
    y = torch.zeros(M, dtype=torch.float32, device=DEV)
    part = torch.zeros(rows * 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_to1_fwd", xa, wdev, bdev, y, part, None, 0, N, D, H, W, C, 27, dtype_code(dt), s)
    check(y.view(N, 1, D, H, W), ref, dt, "to1 fwd (brick)", out_rounded=False)
    st = back(part).view(-1, 2).sum(0)
    assert abs(st[0].item() - ref.sum().item()) <= 1e-4 * max(1.0, ref.abs().sum().item())
    assert abs(st[1].item() - (ref * ref).sum().item()) <= 1e-4 * max(1.0, (ref * ref).sum().item())
    # with a workspace the weight tiles are packed once per call and (C <= 128) the halo is staged by LDS-DMA into two chunk buffers
    # (to1_brick_dma_kernel): the same MFMAs in the same order -- bit-identical output and statistics rows
    nbf = L.call("pcrl_conv3d_to1_fwd_ws_bytes", N, D, H, W, C, 27)
    yw = torch.zeros(M, dtype=torch.float32, device=DEV)
    partw = torch.zeros(rows * 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_to1_fwd", xa, wdev, bdev, yw, partw, ops.workspace(nbf, xa.device), nbf, N, D, H, W, C, 27, dtype_code(dt), s)
    torch.cuda.synchronize()
    assert torch.equal(yw, y) and torch.equal(partw, part)
    # the two-pass kernel (debug switch) rounds at the same points: float32 sums of bf16 products, different order only
    L.debug_set_conv_impl(1)
    try:
        nbf = L.call("pcrl_conv3d_to1_fwd_ws_bytes", N, D, H, W, C, 27)
        y2 = torch.zeros(M, dtype=torch.float32, device=DEV)
        L.call("pcrl_conv3d_to1_fwd", xa, wdev, bdev, y2, None, ops.workspace(nbf, xa.device), nbf, N, D, H, W, C, 27, dtype_code(dt), s)
    finally:
        L.debug_set_conv_impl(0)
    assert (y - y2).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,taps", [(64, 27), (128, 27), (256, 27), (64, 1)])
def test_conv_to_one_channel(C, taps, dt):
    N, D, H, W = 2, 5, 6, 4
    L, s = lib(), stream_handle()
    k = 3 if taps == 27 else 1
    x, w, b = rnd(N, C, D, H, W, seed=1), rnd(1, C, k, k, k, seed=2, scale=0.2), rnd(1, seed=3)
    xq = q(x, dt).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = F.conv3d(xq, wr, br, padding=k // 2)
    M = N * D * H * W
    xa = act_dev(x, dt)
    y = torch.zeros(M, dtype=torch.float32, device=DEV)
    part = torch.zeros(((M + 1023) // 1024) * 2, dtype=torch.float32, device=DEV)
    wdev, bdev = w.float().to(DEV), b.float().to(DEV)
    # direct gather kernel (no workspace) and the two-pass pointwise-GEMM + shifted-sum path (27 taps, with workspace)
    L.call("pcrl_conv3d_to1_fwd", xa, wdev, bdev, y, part, None, 0, N, D, H, W, C, taps, dtype_code(dt), s)
    check(y.view(N, 1, D, H, W), ref, dt, "to1 fwd (gather)", out_rounded=False)
    st = back(part).view(-1, 2).sum(0)
    assert abs(st[0].item() - ref.sum().item()) <= 1e-4 * max(1.0, ref.abs().sum().item())
    nbf = L.call("pcrl_conv3d_to1_fwd_ws_bytes", N, D, H, W, C, taps)
    if nbf:
        y.zero_()
        part.zero_()
        L.call("pcrl_conv3d_to1_fwd", xa, wdev, bdev, y, part, ops.workspace(nbf, xa.device), nbf, N, D, H, W, C, taps, dtype_code(dt), s)
        # bf16: the tap weights are MFMA operands here (rounded to bf16)
        ref2 = F.conv3d(xq.detach(), q(w, dt), b, padding=k // 2) if dt == torch.bfloat16 else ref
        check(y.view(N, 1, D, H, W), ref2, dt, "to1 fwd (pointwise GEMM + shifted sum)", out_rounded=False)
        st = back(part).view(-1, 2).sum(0)
        assert abs(st[0].item() - ref2.sum().item()) <= 1e-4 * max(1.0, ref2.abs().sum().item())
        assert abs(st[1].item() - (ref2 * ref2).sum().item()) <= 1e-4 * max(1.0, (ref2 * ref2).sum().item())
    dy = rnd(N, 1, D, H, W, seed=4)
    ref.backward(dy)
    gx = xq.grad.clone()
    # weight gradient: MFMA GEMM on x and the im2col of dy, both stored in the activation dtype (27 taps); the 1x1x1 case is a weighted
    # column sum of x with the float32 dy (no rounding of dy)
    wr.grad = None
    F.conv3d(xq.detach(), wr, None, padding=k // 2).backward(q(dy, dt) if taps == 27 else dy)
    dyd = dy.float().to(DEV).contiguous()
    add = rnd(N, C, D, H, W, seed=6)
    adda = act_dev(add, dt)
    dx = ops.new_act(N, D, H, W, C, dt, DEV)
    L.call("pcrl_conv3d_to1_dgrad", dyd, wdev, adda, dx, N, D, H, W, C, taps, dtype_code(dt), s)
    check(dx, gx + q(add, dt), dt, "to1 dgrad (+add_src)")
    L.call("pcrl_conv3d_to1_dgrad", dyd, wdev, None, dx, N, D, H, W, C, taps, dtype_code(dt), s)
    check(dx, gx, dt, "to1 dgrad")
    nb = L.call("pcrl_conv3d_to1_wgrad_ws_bytes", N, D, H, W, C, taps)
    dw = torch.zeros_like(wdev)
    db = torch.zeros(1, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_to1_wgrad", xa, dyd, dw, db, ops.workspace(nb, xa.device), nb, N, D, H, W, C, taps, dtype_code(dt), s)
    check(dw, wr.grad, dt, "to1 wgrad", out_rounded=False, f32_tol=3e-5)
    check(db, br.grad, dt, "to1 bias grad", out_rounded=False, f32_tol=3e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("act", [ACT_RELU, ACT_NONE])
@pytest.mark.parametrize("shape", [(3, 6, 5, 7, 32), (2, 4, 4, 4, 512)])
def test_batchnorm_act(shape, act, dt):
    """luconv-style BN: statistics from partials, apply, backward (vs torch batch_norm autograd)."""
    N, D, H, W, C = shape
    M = N * D * H * W
    y = rnd(N, C, D, H, W, seed=1, scale=2.0) + rnd(1, C, 1, 1, 1, seed=9)
    gamma, beta = 1 + 0.3 * rnd(C, seed=2), 0.3 * rnd(C, seed=3)
    yq = q(y, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    z = F.batch_norm(yq, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    a = torch.relu(z) if act == ACT_RELU else z
    da = rnd(N, C, D, H, W, seed=4)
    a.backward(q(da, dt))
    # device: partial statistics as the conv epilogue would emit them (computed here from the rounded tensor)
    ya = act_dev(y, dt)
    flat = back(ya).permute(0, 2, 3, 4, 1).reshape(M, C)
    rows = (M + CONV_BM - 1) // CONV_BM
    pad = torch.zeros(rows * CONV_BM - M, C, dtype=torch.float64)
    fp = torch.cat([flat, pad]).view(rows, CONV_BM, C)
    part = torch.stack([fp.sum(1), (fp * fp).sum(1)], dim=-1).float().to(DEV).contiguous()
    g32, b32 = gamma.float().to(DEV), beta.float().to(DEV)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, rstd, scale, shift = ops.bn_finalize(part.view(-1), rows, C, M, g32, b32, rmd, rvd)
    check(rmd, rm, dt, "running_mean", out_rounded=False, f32_tol=1e-5)
    check(rvd, rv, dt, "running_var", out_rounded=False, f32_tol=1e-5)
    out = ops.bn_act_apply(ya, scale, shift, M, C, act, dt)
    check(out, a.detach(), dt, "bn apply")
    dy, dg, dbeta = ops.bn_act_backward(act_dev(da, dt), ya, g32, mean, rstd, scale, shift, M, C, act, dt)
    check(dy, yq.grad, dt, "bn bwd dx")
    check(dg, gr.grad, dt, "bn dgamma", out_rounded=False, f32_tol=1e-4)
    check(dbeta, br.grad, dt, "bn dbeta", out_rounded=False, f32_tol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("with_da", [True, False])
@pytest.mark.parametrize("shape", [(3, 8, 8, 16, 64), (2, 4, 4, 4, 256), (5, 2, 2, 2, 32), (2, 6, 5, 7, 32)])
def test_batchnorm_backward_with_global_average_pool_branch_folded_in(shape, with_da, dt):
    """UpTransition.forward (pcrlv2_model_3d.py:64-70): a1 = relu(bn(conv)) feeds the next stage AND adaptive_avg_pool3d.  autograd's
    gradient of a1 is d_out + d_g[n][c] / S; pcrl_bn_act_bwd_*_rowadd takes the pool branch as a [N][C] term inside both passes of the
    BatchNorm backward (d_out may be absent: passes whose reconstruction output is unused).  Reference: torch float64 autograd of
    relu(batch_norm(y)) with gradients arriving through both consumers.  (tile divides the sample / several samples per tile / rows
    that are no multiple of anything all occur in the shapes.)"""
    N, D, H, W, C = shape
    S, M = D * H * W, N * D * H * W
    assert ops.bn_rowadd_ok(C, dt)
    y = rnd(N, C, D, H, W, seed=1, scale=2.0) + rnd(1, C, 1, 1, 1, seed=9)
    gamma, beta = 1 + 0.3 * rnd(C, seed=2), 0.3 * rnd(C, seed=3)
    yq = q(y, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = torch.relu(F.batch_norm(yq, None, None, gr, br, training=True, eps=1e-5))
    da = q(rnd(N, C, D, H, W, seed=4), dt)
    dg = rnd(N, C, seed=5, scale=float(S) ** 0.5).float().double()     # float32 on the device: use the same values here
    loss = (a.mean(dim=(2, 3, 4)) * dg).sum()
    if with_da:
        loss = loss + (a * da).sum()
    loss.backward()
    ya = act_dev(y, dt)
    flat = back(ya).permute(0, 2, 3, 4, 1).reshape(M, C)
    rows = (M + CONV_BM - 1) // CONV_BM
    fp = torch.cat([flat, torch.zeros(rows * CONV_BM - M, C, dtype=torch.float64)]).view(rows, CONV_BM, C)
    part = torch.stack([fp.sum(1), (fp * fp).sum(1)], dim=-1).float().to(DEV).contiguous()
    g32, b32 = gamma.float().to(DEV), beta.float().to(DEV)
    mean, rstd, scale, shift = ops.bn_finalize(part.view(-1), rows, C, M, g32, b32, torch.zeros(C, device=DEV), torch.ones(C, device=DEV))
    dy, dgam, dbeta = ops.bn_act_backward(act_dev(da, dt) if with_da else None, ya, g32, mean, rstd, scale, shift, M, C, ACT_RELU, dt,
                                          row_g=dg.float().to(DEV).contiguous())
    check(dy, yq.grad, dt, "bn bwd dx (row term)")
    check(dgam, gr.grad, dt, "bn dgamma (row term)", out_rounded=False, f32_tol=1e-4)
    check(dbeta, br.grad, dt, "bn dbeta (row term)", out_rounded=False, f32_tol=1e-4)
    # forward side of the same fusion: activation + global average pool in one pass (pcrl_bn_act_apply_gap)
    a_ref = ops.bn_act_apply(ya, scale, shift, M, C, ACT_RELU, dt)
    a1, g1 = torch.empty_like(ya), torch.empty(N, C, dtype=torch.float32, device=DEV)
    nb = lib().call("pcrl_gap_ws_bytes", N, S, C)
    lib().call("pcrl_bn_act_apply_gap", ya, a1, g1, scale, shift, ops.workspace(nb, DEV_T), nb, N, S, C, ACT_RELU, dtype_code(dt), stream_handle())
    assert torch.equal(a1, a_ref)
    check(g1, back(a_ref).mean(dim=(2, 3, 4)), dt, "fused global average pool", out_rounded=False, f32_tol=1e-5)
    # and the materialised form it replaces: same result up to the one bf16 rounding of the materialised sum
    dsum = ops.gap_backward(dg.float().to(DEV).contiguous(), ya, act_dev(da, dt) if with_da else None, dt)
    dy2, dgam2, dbeta2 = ops.bn_act_backward(dsum, ya, g32, mean, rstd, scale, shift, M, C, ACT_RELU, dt)
    check(dy, back(dy2), dt, "row term vs materialised", bf_tol=2e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("shape", [(2, 8, 8, 16, 64), (3, 2, 4, 6, 32), (1, 4, 4, 4, 256)])
def test_batchnorm_backward_with_maxpool_backward_folded_in(shape, ties, dt):
    """Encoder stage end (pcrlv2_model_3d.py:115-117): a = relu(bn(y)) is consumed by nn.MaxPool3d(2) only.  pcrl_bn_act_bwd_*_pool take
    the POOLED tensor's gradient and fold max_pool3d_backward into both BatchNorm passes.  Checked against (1) torch float64 autograd of
    max_pool3d(relu(batch_norm(y))) where no ties exist, and (2) the device's own three-kernel route (bn_act_apply -> maxpool_backward ->
    bn_act_backward) on inputs full of ties -- equal positive values inside a window and whole windows of zeros -- where the first
    maximum in scan order must take the gradient."""
    N, D, H, W, C = shape
    M = N * D * H * W
    assert ops.bn_pool_ok(D, H, W, C, dt)
    y = rnd(N, C, D, H, W, seed=1, scale=2.0) + rnd(1, C, 1, 1, 1, seed=9)
    if ties:
        y = (y * 2).round() / 2          # a handful of distinct values: every window has repeated maxima
    gamma, beta = 1 + 0.3 * rnd(C, seed=2), 0.3 * rnd(C, seed=3)
    dp = rnd(N, C, D // 2, H // 2, W // 2, seed=4)
    ya = act_dev(y, dt)
    flat = back(ya).permute(0, 2, 3, 4, 1).reshape(M, C)
    rows = (M + CONV_BM - 1) // CONV_BM
    fp = torch.cat([flat, torch.zeros(rows * CONV_BM - M, C, dtype=torch.float64)]).view(rows, CONV_BM, C)
    part = torch.stack([fp.sum(1), (fp * fp).sum(1)], dim=-1).float().to(DEV).contiguous()
    g32, b32 = gamma.float().to(DEV), beta.float().to(DEV)
    mean, rstd, scale, shift = ops.bn_finalize(part.view(-1), rows, C, M, g32, b32, torch.zeros(C, device=DEV), torch.ones(C, device=DEV))
    dpa = act_dev(dp, dt)
    dy, dgam, dbeta = ops.bn_act_backward(None, ya, g32, mean, rstd, scale, shift, M, C, ACT_RELU, dt, pool_dp=dpa)
    # the three-kernel route on the same inputs
    a = ops.bn_act_apply(ya, scale, shift, M, C, ACT_RELU, dt)
    # forward pair in one pass (pcrl_bn_act_apply_pool): bit-identical to apply + pool
    a1, p1 = torch.empty_like(ya), ops.new_act(N, D // 2, H // 2, W // 2, C, dt, DEV_T)
    lib().call("pcrl_bn_act_apply_pool", ya, a1, p1, scale, shift, N, D, H, W, C, ACT_RELU, dtype_code(dt), stream_handle())
    assert torch.equal(a1, a) and torch.equal(p1, ops.maxpool_forward(a, dt))
    dfull = ops.maxpool_backward(a, dpa, dt)
    dy2, dgam2, dbeta2 = ops.bn_act_backward(dfull, ya, g32, mean, rstd, scale, shift, M, C, ACT_RELU, dt)
    check(dgam, back(dgam2), dt, "dgamma vs three kernels", out_rounded=False, f32_tol=2e-5)
    check(dbeta, back(dbeta2), dt, "dbeta vs three kernels", out_rounded=False, f32_tol=2e-5)
    check(dy, back(dy2), dt, "dy vs three kernels", f32_tol=2e-5, bf_tol=8e-3)
    if not ties:
        yq = q(y, dt).requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        pooled = F.max_pool3d(torch.relu(F.batch_norm(yq, None, None, gr, br, training=True, eps=1e-5)), 2)
        pooled.backward(q(dp, dt))
        if dt == torch.float32:     # in bf16 the device's argmax is taken on the ROUNDED activation: near-ties may pick another voxel
            check(dy, yq.grad, dt, "dy vs torch")
        check(dgam, gr.grad, dt, "dgamma vs torch", out_rounded=False, f32_tol=1e-4 if dt == torch.float32 else 2e-2)
        check(dbeta, br.grad, dt, "dbeta vs torch", out_rounded=False, f32_tol=1e-4 if dt == torch.float32 else 2e-2)


def test_batchnorm_one_channel_sigmoid():
    N, D, H, W = 2, 8, 8, 4
    M = N * D * H * W
    y = rnd(N, 1, D, H, W, seed=1, scale=2.0) + 0.3
    yr = y.clone().requires_grad_(True)
    g, b = torch.tensor([1.2], dtype=torch.float64, requires_grad=True), torch.tensor([-0.1], dtype=torch.float64, requires_grad=True)
    a = torch.sigmoid(F.batch_norm(yr, None, None, g, b, training=True, eps=1e-5))
    da = rnd(N, 1, D, H, W, seed=2)
    a.backward(da)
    yd = y.float().to(DEV).view(-1)
    part = torch.stack([yd.double().sum(), (yd.double() ** 2).sum()]).float().view(1, 2).contiguous()
    mean, rstd, scale, shift = ops.bn_finalize(part.view(-1), 1, 1, M, g.detach().float().to(DEV), b.detach().float().to(DEV), None, None)
    out = ops.bn_act_apply(yd, scale, shift, M, 1, ACT_SIGMOID, torch.float32)
    check(out.view(N, 1, D, H, W), a.detach(), torch.float32, "bn1 sigmoid apply")
    dy, dg, dbeta = ops.bn_act_backward(da.float().to(DEV).view(-1), yd, g.detach().float().to(DEV), mean, rstd, scale, shift, M, 1, ACT_SIGMOID, torch.float32)
    check(dy.view(N, 1, D, H, W), yr.grad, torch.float32, "bn1 sigmoid bwd", f32_tol=1e-4)
    check(dg, g.grad, torch.float32, "bn1 dgamma", f32_tol=1e-4)
    check(dbeta, b.grad, torch.float32, "bn1 dbeta", f32_tol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
def test_maxpool_with_ties(dt):
    N, C, D, H, W = 2, 64, 4, 6, 4
    x = torch.round(rnd(N, C, D, H, W, seed=1) * 2) / 2  # few distinct values -> many ties
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool3d(xr, 2)
    dy = rnd(N, C, D // 2, H // 2, W // 2, seed=2)
    ref.backward(q(dy, dt))
    xa = act_dev(x, dt)
    y = ops.maxpool_forward(xa, dt)
    check(y, ref.detach(), dt, "maxpool fwd", bf_tol=1e-6)
    dx = ops.maxpool_backward(xa, act_dev(dy, dt), dt)
    check(dx, xr.grad, dt, "maxpool bwd (first max takes the gradient)", bf_tol=1e-6)


@pytest.mark.parametrize("dt", DTYPES)
def test_gap_and_colsum(dt):
    N, C, D, H, W = 3, 128, 12, 10, 9  # S = 1080 > one 1024-row tile
    x = rnd(N, C, D, H, W, seed=1)
    xa = act_dev(x, dt)
    g = ops.gap_forward(xa, dt)
    check(g, q(x, dt).mean(dim=(2, 3, 4)), dt, "gap fwd", out_rounded=False)
    dg = rnd(N, C, seed=2)
    add = rnd(N, C, D, H, W, seed=3)
    da = ops.gap_backward(dg.float().to(DEV), xa, act_dev(add, dt), dt)
    check(da, q(add, dt) + (dg / (D * H * W)).view(N, C, 1, 1, 1), dt, "gap bwd + add_src")
    da = ops.gap_backward(dg.float().to(DEV), xa, None, dt)
    check(da, (dg / (D * H * W)).view(N, C, 1, 1, 1).expand(N, C, D, H, W), dt, "gap bwd")
    L = lib()
    M = N * D * H * W
    nb = L.call("pcrl_colsum_ws_bytes", M, C)
    out = torch.zeros(C, dtype=torch.float32, device=DEV)
    L.call("pcrl_colsum", xa, out, ops.workspace(nb, xa.device), nb, M, C, dtype_code(dt), stream_handle())
    check(out, q(x, dt).sum(dim=(0, 2, 3, 4)), dt, "colsum", out_rounded=False)


@pytest.mark.parametrize("rows,C", [(4, 64), (32, 256), (24, 128)])
def test_heads_bn1d_linear(rows, C):
    dt = torch.float32
    x = rnd(rows, C, seed=1).requires_grad_(True)
    g, b = (1 + 0.2 * rnd(C, seed=2)).requires_grad_(True), (0.2 * rnd(C, seed=3)).requires_grad_(True)
    for relu in (False, True):
        for t in (x, g, b):
            t.grad = None
        rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
        z = F.batch_norm(x, rm, rv, g, b, training=True, momentum=0.1, eps=1e-5)
        y = torch.relu(z) if relu else z
        dy = rnd(rows, C, seed=4)
        y.backward(dy)
        rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        xd = x.detach().float().to(DEV)
        yd, mean, rstd = ops.bn1d_forward(xd, g.detach().float().to(DEV), b.detach().float().to(DEV), rmd, rvd, relu)
        check(yd, y.detach(), dt, "bn1d fwd", f32_tol=1e-5)
        check(rvd, rv, dt, "bn1d running_var", f32_tol=1e-5)
        check(rmd, rm, dt, "bn1d running_mean", f32_tol=1e-5)
        dx, dg, db = ops.bn1d_backward(dy.float().to(DEV), xd, yd, g.detach().float().to(DEV), mean, rstd, relu)
        check(dx, x.grad, dt, "bn1d bwd", f32_tol=2e-4)
        check(dg, g.grad, dt, "bn1d dgamma", f32_tol=1e-4)
        check(db, b.grad, dt, "bn1d dbeta", f32_tol=1e-4)
    w, bias = rnd(2 * C, C, seed=5, scale=0.1).requires_grad_(True), rnd(2 * C, seed=6).requires_grad_(True)
    x.grad = None
    yl = F.linear(x, w, bias)
    dyl = rnd(rows, 2 * C, seed=7)
    yl.backward(dyl)
    xd, wd, bd = x.detach().float().to(DEV), w.detach().float().to(DEV), bias.detach().float().to(DEV)
    check(ops.linear_forward(xd, wd, bd), yl.detach(), dt, "linear fwd")
    dx, dw, db = ops.linear_backward(dyl.float().to(DEV), xd, wd)
    check(dx, x.grad, dt, "linear dx")
    check(dw, w.grad, dt, "linear dw")
    check(db, bias.grad, dt, "linear db")
    with pytest.raises(RuntimeError, match="more than 1 value per channel"):
        ops.bn1d_forward(xd[:1].contiguous(), wd[0].contiguous(), wd[1].contiguous(), None, None, False)


@pytest.mark.parametrize("shape", [(2, 8, 16, 32, 64, 64), (1, 4, 8, 16, 32, 96), (3, 4, 16, 8, 64, 64), (1, 16, 16, 8, 128, 32), (2, 8, 16, 24, 32, 96),
                                   (2, 8, 8, 4, 64, 128), (2, 4, 8, 8, 64, 64)])
def test_brick_convolution_kernels_equal_the_gather_kernel_to_an_ulp(shape):
    """bf16 3x3x3 convolution: the wide-brick kernel (bricks along (D, H, W) and along (D, W, H)) and the 4x8x8-brick kernel against the gather
    kernel (`pcrl_debug_set_conv_impl(1)`, the one the exact-float32 mode pins against float64) on the same operands -- the same bf16 products
    summed in float32 in another order: outputs differ by at most one bf16 ulp on a fraction of a percent of the elements, the float32
    statistics rows agree to 1e-5.  A wrong tap or border anywhere in a brick kernel is an O(1) difference here."""
    N, D, H, W, Ci, Co = shape
    dt = torch.bfloat16
    L, s = lib(), stream_handle()
    g = torch.Generator(device=DEV).manual_seed(9)
    x = ops.new_act(N, D, H, W, Ci, dt, DEV)
    x.normal_(generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, device=DEV, generator=g) * 0.05
    b = torch.randn(Co, device=DEV, generator=g) * 0.2
    wf, _ = ops.PackedWeights("conv3").get(w, dt)
    outs, kinds = [], []
    try:
        for impl in (0, 1):
            L.debug_set_conv_impl(impl)
            code = dtype_code(dt)
            kinds.append(L.call("pcrl_conv3d_k3_fwd_kernel", N, D, H, W, Ci, Co, code))
            rows = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, code)
            y = ops.new_act(N, D, H, W, Co, dt, DEV)
            part = torch.zeros(rows, Co, 2, device=DEV)
            nb = L.call("pcrl_conv3d_k3_fwd_ws_bytes", N, D, H, W, Ci, Co, code)
            L.call("pcrl_conv3d_k3_fwd_ws", x, wf, b, y, part, ops.workspace(nb, torch.device(DEV)) if nb else None, nb, N, D, H, W, Ci, Co, code, s)
            outs.append((y.float(), part.double().sum(0)))
    finally:
        L.debug_set_conv_impl(0)
    assert kinds[0] in (1, 2) and kinds[1] == 0, kinds
    (ya, pa), (yb, pb) = outs
    d = (ya - yb).abs()
    assert float((d > 0).float().mean()) < 5e-3 and float((d / yb.abs().clamp_min(1e-2)).max()) <= 2.0 ** -7 + 1e-6, (kinds, float(d.max()))
    # (sum, sum of squares) per channel: float32 partial sums in another order -- 1e-6 of sum|y| <= sqrt(count * sum y^2), resp. of sum y^2
    s2 = pb[:, 1]
    assert bool(((pa[:, 0] - pb[:, 0]).abs() <= 1e-6 * torch.sqrt(N * D * H * W * s2)).all()) and bool(((pa[:, 1] - s2).abs() <= 1e-6 * s2).all())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ratio", [10.0, 100.0])
def test_batch_statistics_on_badly_centred_channels(ratio, dt):
    """Training-mode BatchNorm statistics come from per-tile (sum, sum of squares) of the float32 accumulators, reduced in float64 -- the
    variance is E[y^2] - mean^2, which cancels when |mean| >> std (aten uses Welford).  Measured here on a convolution whose
    bias puts every channel's mean at `ratio` standard deviations: the per-tile float32 rounding is unbiased and averages out over the tiles
    (512-voxel bricks / 128-row tiles), so rstd stays within 2e-5 (ratio 10; measured 3e-6) / 2e-3 (ratio 100; measured 3.5e-4) of the float64 value -- far inside what the
    activations' own rounding moves -- instead of the worst case (mean/std)^2 * 1e-7 per tile times no averaging."""
    N, D, H, W, Ci, Co = 2, 8, 16, 32, 32, 64
    x = rnd(N, Ci, D, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, 3, seed=2, scale=0.05)
    y0 = F.conv3d(q(x, dt), q(w, dt), None, padding=1)
    std = y0.std(dim=(0, 2, 3, 4))
    bias = ratio * std * (1 + 0.1 * rnd(Co, seed=3))
    y = y0 + bias.view(1, -1, 1, 1, 1)
    mean_ref, var_ref = y.mean(dim=(0, 2, 3, 4)), y.var(dim=(0, 2, 3, 4), unbiased=False)
    g, b = torch.ones(Co), torch.zeros(Co)
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    packed = ops.PackedWeights("conv3")
    a, sv = ops.luconv_forward(ops.to_act(x.to(DEV), dt), w.float().to(DEV), bias.float().to(DEV), g.to(DEV), b.to(DEV), rm, rv, packed, 0, dt)
    rstd_ref = 1.0 / torch.sqrt(var_ref + 1e-5)
    e_mean = float(((back(sv.mean) - mean_ref).abs() / std).max())
    e_rstd = float(((back(sv.rstd) - rstd_ref).abs() / rstd_ref).max())
    print(f"  mean/std = {ratio:g} [{dt}]: max |mean - ref| / std = {e_mean:.2e}, max relative rstd error = {e_rstd:.2e}")
    assert e_mean < 1e-4 and e_rstd < (2e-5 if ratio <= 10 else 2e-3)


@pytest.mark.parametrize("rows,Cin,Cout", [(32, 64, 128), (32, 512, 256), (192, 256, 512), (192, 128, 64), (5, 36, 10), (33, 260, 7), (7, 30, 12)])
def test_linear_products(rows, Cin, Cout):
    """pcrl_linear_fwd / _bwd (nn.Linear of the predictor heads, pcrlv2_model_3d.py:57-58): y = x W^T + b, dx = dy W, dW = dy^T x, db = colsum(dy)
    against float64.  Head sizes and ragged ones (partial row groups, channel counts not a multiple of 64 or 4 -- the last takes the tiled
    fallback kernel), contiguous float32 operands."""
    x, w, b = rnd(rows, Cin, seed=1), rnd(Cout, Cin, seed=2, scale=0.1), rnd(Cout, seed=3)
    dy = rnd(rows, Cout, seed=4)
    xd, wd, bd, dyd = (t.float().to(DEV) for t in (x, w, b, dy))
    x64, w64, dy64 = (t.float().double() for t in (x, w, dy))
    check(ops.linear_forward(xd, wd, bd), x64 @ w64.T + b.float().double(), torch.float32, "linear fwd", f32_tol=2e-6)
    dx, dw, db = ops.linear_backward(dyd, xd, wd)
    check(dx, dy64 @ w64, torch.float32, "linear dx", f32_tol=2e-6)
    check(dw, dy64.T @ x64, torch.float32, "linear dw", f32_tol=2e-6)
    check(db, dy64.sum(0), torch.float32, "linear db", f32_tol=2e-6)


@pytest.mark.parametrize("scale", [2, 4])
def test_trilinear(scale):
    N, D, H, W = 2, 4, 6, 3
    x = rnd(N, 1, D, H, W, seed=1).requires_grad_(True)
    ref = F.interpolate(x, scale_factor=scale, mode="trilinear")
    dy = rnd(*ref.shape, seed=2)
    ref.backward(dy)
    y = ops.upsample_forward(x.detach().float().to(DEV), scale)
    check(y, ref.detach(), torch.float32, "trilinear fwd")
    dx = ops.upsample_backward(dy.float().to(DEV), tuple(x.shape), scale)
    check(dx, x.grad, torch.float32, "trilinear bwd")
    probe = ops.upsample_forward(torch.arange(4, dtype=torch.float32, device=DEV).view(1, 1, 1, 1, 4), 2)
    np.testing.assert_allclose(back(probe)[0, 0, 0, 0].numpy(), [0, .25, .75, 1.25, 1.75, 2.25, 2.75, 3], atol=1e-6)


def test_losses_and_sigmoid():
    p, gt = torch.sigmoid(rnd(2, 1, 8, 8, 9, seed=1)).requires_grad_(True), rnd(2, 1, 8, 8, 9, seed=2).abs()
    ref = F.mse_loss(p, gt)
    ref.backward(torch.tensor(0.7, dtype=torch.float64))
    pd, gd = p.detach().float().to(DEV), gt.float().to(DEV)
    check(ops.mse_forward(pd, gd).view(1), ref.detach().view(1), torch.float32, "mse fwd")
    check(ops.mse_backward(pd, gd, torch.tensor(0.7, device=DEV)), p.grad, torch.float32, "mse bwd")
    for rows, C in ((4, 64), (32, 256)):
        x, y = rnd(rows, C, seed=3).requires_grad_(True), rnd(rows, C, seed=4)
        c = torch.nn.CosineSimilarity()(x, y).mean()
        c.backward(torch.tensor(-0.5, dtype=torch.float64))
        xd, yd = x.detach().float().to(DEV), y.float().to(DEV)
        out, saved = ops.cosine_mean_forward(xd, yd)
        check(out.view(1), c.detach().view(1), torch.float32, "cosine fwd")
        check(ops.cosine_mean_backward(xd, yd, saved, torch.tensor(-0.5, device=DEV)), x.grad, torch.float32, "cosine bwd")


def test_fused_sgd_matches_torch_sgd():
    from pcrlv2_amd.optim import FusedSGD
    torch.manual_seed(0)
    shapes = [(32, 1, 3, 3, 3), (32,), (7,), (64, 32, 3, 3, 3), (5, 3)]
    ref_p = [torch.randn(s, dtype=torch.float64, requires_grad=True) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().float().to(DEV)) for p in ref_p]
    ref = torch.optim.SGD(ref_p, lr=0.05, momentum=0.9, weight_decay=1e-2)
    mine = FusedSGD(my_p, lr=0.05, momentum="0.9", weight_decay="1e-2")   # strings, like the reference CLI can pass
    for step in range(4):
        for i, (a, b) in enumerate(zip(ref_p, my_p)):
            if i == 2 and step in (0, 2):   # parameter without a gradient in some steps (deep-supervision heads)
                a.grad, b.grad = None, None
                continue
            g = torch.randn(a.shape, dtype=torch.float64)
            a.grad, b.grad = g, g.float().to(DEV)
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        check(b.detach(), a.detach(), torch.float32, "sgd param")
    sd = mine.state_dict()
    assert set(sd["state"][0].keys()) == {"momentum_buffer"} and sd["param_groups"][0]["momentum"] == 0.9


def test_sgd_kernel_on_an_unpadded_arena():
    """pcrl_sgd_step on a flat arena whose tensors start anywhere (FusedSGD pads its slots to 4 floats; a caller of the C ABI need not): groups
    of four elements that straddle two tensors -- with different has-gradient / has-momentum flags -- take the scalar path of the kernel."""
    L, s = lib(), stream_handle()
    sizes = [5, 1, 7, 64, 3, 130, 2]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    total = offs[-1]
    flags = [1, 3, 0, 3, 1, 2, 3]          # bit 0: has a gradient, bit 1: momentum buffer initialised
    g = torch.Generator().manual_seed(4)
    p0, gr, b0 = (torch.randn(total, generator=g, dtype=torch.float64) for _ in range(3))
    lr, mom, wd, gs = 0.05, 0.9, 1e-2, 0.5
    ep, eb = p0.clone(), b0.clone()
    for t, (o, n) in enumerate(zip(offs, sizes)):
        if not flags[t] & 1:
            continue
        sl = slice(o, o + n)
        gg = gr[sl] * gs + wd * p0[sl]
        bb = mom * b0[sl] + gg if flags[t] & 2 else gg
        eb[sl] = bb
        ep[sl] = p0[sl] - lr * bb
    pd, gd, bd = (t.float().to(DEV) for t in (p0, gr, b0))
    L.call("pcrl_sgd_step", pd, gd, bd, torch.tensor(offs, dtype=torch.int64, device=DEV), torch.tensor(flags, dtype=torch.int32, device=DEV),
           len(sizes), total, lr, mom, wd, gs, s)
    check(pd, ep, torch.float32, "sgd parameters", f32_tol=2e-6)
    check(bd, eb, torch.float32, "sgd momentum buffers", f32_tol=2e-6)


@pytest.mark.parametrize("N,C,tau", [(32, 64, 0.5), (96, 256, 0.1), (2, 8, 1.0)])
def test_ntxent_optional_extra(N, C, tau):
    """NT-Xent (named in north_star, absent from the reference) against its PyTorch float64 definition:
    cross_entropy(zn zn^T / tau with -inf diagonal, target = the other view), value and gradient w.r.t. both views."""
    from pcrlv2_amd.functions import ntxent_loss
    z1, z2 = rnd(N, C, seed=51), rnd(N, C, seed=52)
    a, b = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
    zn = F.normalize(torch.cat([a, b]), dim=1, eps=1e-8)
    sim = zn @ zn.T / tau
    sim = sim.masked_fill(torch.eye(2 * N, dtype=torch.bool), float("-inf"))
    ref = F.cross_entropy(sim, (torch.arange(2 * N) + N) % (2 * N))
    ref.backward()
    x1, x2 = z1.float().to(DEV).detach().requires_grad_(True), z2.float().to(DEV).detach().requires_grad_(True)
    loss = ntxent_loss(x1, x2, tau)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    for g, r in ((x1.grad, a.grad), (x2.grad, b.grad)):
        d = (g.double().cpu() - r).abs().max().item()
        assert d <= 2e-5 * max(1e-3, r.abs().max().item()) + 1e-8, d


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(3, 64, 4, 6, 5), (2, 32, 8, 8, 8), (2, 128, 2, 3, 4)])
def test_groupnorm_silu_optional_extra(shape, dt):
    """GroupNorm(8)+SiLU (named in north_star, absent from the reference) against torch's
    F.silu(F.group_norm(...)) in float64: output, input gradient, dgamma, dbeta."""
    from pcrlv2_amd.functions import group_norm_silu
    N, C, D, H, W = shape
    x, g, b = rnd(N, C, D, H, W, seed=61), rnd(C, seed=62) * 0.5 + 1.0, rnd(C, seed=63) * 0.1
    dy = rnd(N, C, D, H, W, seed=64)
    xq = q(x, dt).double().requires_grad_(True)
    gr, br = g.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)
    ref = F.silu(F.group_norm(xq, 8, gr, br, eps=1e-5))
    ref.backward(q(dy, dt).double())
    xa = act_dev(x, dt).detach().requires_grad_(True)
    gd, bd = g.detach().float().to(DEV).requires_grad_(True), b.detach().float().to(DEV).requires_grad_(True)
    out = group_norm_silu(xa, gd, bd, 8)
    check(out, ref.detach(), dt, "gn+silu fwd")
    out.backward(act_dev(dy, dt))
    check(xa.grad, xq.grad, dt, "gn+silu dx")
    check(gd.grad, gr.grad, dt, "gn+silu dgamma", out_rounded=False, f32_tol=1e-4)
    check(bd.grad, br.grad, dt, "gn+silu dbeta", out_rounded=False, f32_tol=1e-4)
