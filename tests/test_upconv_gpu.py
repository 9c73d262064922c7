"""GPU parity of the composed ConvTranspose3d(k2,s2) -> Conv3d(3x3x3) operator (csrc/upconv_fused.hip; UpTransition.forward's
`self.ops(self.up_conv(x))`, pcrlv2_model_3d.py:64) against torch float64 autograd of the TWO aten calls it replaces.

The composed operator never forms the upsampled tensor, so there is no operand to pre-round on both sides:
  float32 mode : exact-fp32 MFMA chains on both factorizations -> 1e-4 * max|ref| (two nested sums, composition included);
  bfloat16 mode: x, dy0 and the COMPOSED weights are rounded to bf16 (the two-call route rounds x, both weights and the upsampled
                 tensor instead) -> 2e-2 * max|ref| on data-sized results, 3e-2 on the weight gradients (they pass through a second
                 bf16 rounding in the chain rule GEMMs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import ACT_RELU, dtype_code, lib, stream_handle  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def back(t):
    torch.cuda.synchronize()      # operator-level tests read results of EVERY engine stream (weight gradients are produced on the side stream, ops.side_wgrad)
    return t.detach().double().cpu().contiguous()


def close(got, ref, tol, what):
    got, ref = back(got), ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max|d|={err:.3e} > {tol:.1e} * {scale:.3e}"


SHAPES = [  # N, D, H, W (coarse), Ci, Cm, Co
    (2, 3, 4, 5, 32, 32, 32),       # odd extents, rows not a multiple of the 128-row tile
    (1, 1, 1, 1, 32, 64, 32),       # a single coarse voxel: every fine voxel is a corner
    (2, 1, 2, 3, 64, 32, 64),       # fine extent 2 along d: first and last class only
    (1, 4, 4, 4, 128, 128, 64),     # up_tr64's channels
    (3, 2, 2, 2, 64, 64, 32),
    (1, 8, 8, 4, 64, 64, 32),       # several row tiles
    # shapes whose composed-weight gradient runs on the brick kernel in bf16 (coarse D % 2 == H % 8 == W % 8 == 0 or the permuted form, Co % 64 == 0):
    (1, 2, 8, 8, 64, 64, 64),       # 64 x 64 tiles, one brick range
    (2, 4, 8, 8, 128, 128, 64),     # 64 x 128 tiles (up_tr64's channels)
    (1, 2, 8, 16, 64, 64, 128),     # 128 x 64 tiles inside a phase
    (1, 8, 8, 2, 64, 64, 64),       # innermost extent 2: permuted brick axes
    (3, 2, 8, 8, 32, 64, 64),       # Ci = 32 tiles
    # shapes whose forward runs on the wide-brick kernel in bf16 (coarse D % 4 == H % 8 == W % 16 == 0, Co % 64 == 0): stage window + fine-voxel scatter
    (1, 4, 8, 16, 64, 64, 64),      # one brick: every fine border class on its faces
    (2, 8, 16, 32, 128, 128, 64),   # up_tr64's channels, several bricks in every direction
    (1, 4, 8, 16, 32, 32, 128),     # two 64-channel tiles per phase
    # the data gradient of the three shapes above with Ci % 64 == 0 also runs on the wide-brick kernel (space-to-depth view of dy0); plus:
    (1, 4, 8, 16, 64, 64, 32),      # Co = 32: one 32-channel chunk per parity (forward on the gather kernel)
    (2, 4, 16, 16, 128, 64, 128),   # Co = 128: four chunks per parity, two 64-channel output tiles
    # wide-brick kernel with its axes along (D, W, H) (coarse H % 16 == 0, W % 8 == 0, not W % 16): phases, parities and border classes permuted
    (1, 4, 16, 8, 64, 64, 64),
    (2, 8, 16, 8, 128, 128, 64),
    (1, 4, 32, 24, 64, 64, 32),
    # coarse grids only the 4 x 8 x 8 brick tiles (conv_brick.hip's composed modes): the local views' 8^3 grid, up_tr256's 8 x 8 x 4 grid (permuted axes)
    (2, 8, 8, 8, 128, 128, 64),
    (1, 8, 8, 4, 64, 64, 64),
    (3, 4, 8, 8, 32, 64, 128),      # two 64-channel tiles per phase; one brick per sample: every border class on its faces
    (1, 16, 8, 4, 64, 32, 32),      # Co = 32: forward on the gather kernel, data gradient on the brick (one chunk per parity, 32-channel output tiles)
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_composed_upconv_matches_the_two_aten_calls(shape, dt):
    N, D, H, W, Ci, Cm, Co = shape
    L, s = lib(), stream_handle()
    bf = dt == torch.bfloat16
    x = rnd(N, Ci, D, H, W, seed=1)
    w_up, b_up = rnd(Ci, Cm, 2, 2, 2, seed=2, scale=0.2), rnd(Cm, seed=3, scale=0.5)
    w0, b0 = rnd(Co, Cm, 3, 3, 3, seed=4, scale=0.1), rnd(Co, seed=5, scale=0.5)
    dy0 = rnd(N, Co, 2 * D, 2 * H, 2 * W, seed=6)
    xq, dyq = x.to(dt).double(), dy0.to(dt).double()
    xr = xq.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w_up, b_up, w0, b0)]
    y_ref = F.conv3d(F.conv_transpose3d(xr, pr[0], pr[1], stride=2), pr[2], pr[3], padding=1)
    y_ref.backward(dyq)

    # device
    to_dev = lambda t: t.float().to(DEV).contiguous()
    wu, bu, wc, bc = (to_dev(t) for t in (w_up, b_up, w0, b0))
    xa = ops.to_act(xq.to(dt).to(DEV), dt)
    comp = ops.ComposedUpConv()
    wf, wd, tab = comp.get(wu, bu, wc, bc, dt, geom=(N, D, H, W))
    rows = L.call("pcrl_upconv_stats_rows", N, D, H, W, Ci, Co, dtype_code(dt))
    y = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, torch.device(DEV))
    part = torch.full((rows, Co, 2), float("nan"), dtype=torch.float32, device=DEV)
    L.call("pcrl_upconv_fwd", xa, wf, comp.w3f, tab, y, part, N, D, H, W, Ci, Co, dtype_code(dt), s)
    close(y, y_ref.detach(), 2e-2 if bf else 1e-4, "y0")
    # statistics rows: (sum, sum^2) of the unrounded accumulators -- against the reference tensor
    got = part.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    ref_s = torch.stack([y_ref.detach().sum((0, 2, 3, 4)), (y_ref.detach() ** 2).sum((0, 2, 3, 4))], dim=1)
    close(got, ref_s, 2e-2 if bf else 1e-4, "statistics")

    dya = ops.to_act(dyq.to(dt).to(DEV), dt)
    dx = ops.new_act(N, D, H, W, Ci, dt, torch.device(DEV))
    # the workspace form (what ops.upconv_luconv_backward calls): small coarse grids split the 64-tap reduction over K (deterministic finish pass)
    nbd = L.call("pcrl_upconv_dgrad_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dt))
    wsd = torch.empty(max(nbd, 16), dtype=torch.uint8, device=DEV)
    L.call("pcrl_upconv_dgrad_ws", dya, wd, comp.wd3, dx, wsd if nbd else None, nbd, N, D, H, W, Ci, Co, dtype_code(dt), s)
    if nbd:     # ... and agrees with the one-pass form to the rounding of its float32 partial sums
        dx1 = torch.empty_like(dx)
        L.call("pcrl_upconv_dgrad", dya, wd, comp.wd3, dx1, N, D, H, W, Ci, Co, dtype_code(dt), s)
        assert (dx.float() - dx1.float()).abs().max().item() <= (2e-2 if dt == torch.bfloat16 else 1e-4) * max(dx1.float().abs().max().item(), 1e-6)
    close(dx, xr.grad, 2e-2 if bf else 1e-4, "dx")

    dwu, dbu, dwc = torch.empty_like(wu), torch.empty_like(bu), torch.empty_like(wc)
    nb = L.call("pcrl_upconv_wgrad_ws_bytes", N, D, H, W, Ci, Cm, Co, dtype_code(dt))
    L.call("pcrl_upconv_wgrad", xa, dya, wu, bu, wc, dwu, dbu, dwc, ops.workspace(nb, torch.device(DEV)), nb, N, D, H, W, Ci, Cm, Co, dtype_code(dt), s)
    close(dwu, pr[0].grad, 3e-2 if bf else 1e-4, "dw_up")
    close(dwc, pr[2].grad, 3e-2 if bf else 1e-4, "dw0")
    close(dbu, pr[1].grad, 1e-2 if bf else 1e-4, "db_up")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_composed_stage_head_through_ops(dt):
    """ops.upconv_luconv_forward / _backward (the pair UpStageFn calls): act(bn(conv1(up_conv(x)))) and every gradient against torch
    float64 autograd with training-mode batch_norm."""
    N, D, H, W, Ci, Cm, Co = 2, 4, 4, 4, 64, 64, 32
    bf = dt == torch.bfloat16
    x = rnd(N, Ci, D, H, W, seed=1)
    w_up, b_up = rnd(Ci, Cm, 2, 2, 2, seed=2, scale=0.2), rnd(Cm, seed=3, scale=0.5)
    w0, b0 = rnd(Co, Cm, 3, 3, 3, seed=4, scale=0.1), rnd(Co, seed=5, scale=0.5)
    gamma, beta = 1 + 0.3 * rnd(Co, seed=7), 0.3 * rnd(Co, seed=8)
    da = rnd(N, Co, 2 * D, 2 * H, 2 * W, seed=6)
    xq, daq = x.to(dt).double(), da.to(dt).double()
    xr = xq.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w_up, b_up, w0, b0, gamma, beta)]
    y = F.conv3d(F.conv_transpose3d(xr, pr[0], pr[1], stride=2), pr[2], pr[3], padding=1)
    a_ref = torch.relu(F.batch_norm(y, None, None, pr[4], pr[5], training=True, eps=1e-5))
    a_ref.backward(daq)

    to_dev = lambda t: t.float().to(DEV).contiguous()
    wu, bu, wc, bc, g32, be32 = (to_dev(t) for t in (w_up, b_up, w0, b0, gamma, beta))
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    comp = ops.ComposedUpConv()
    a, sv = ops.upconv_luconv_forward(ops.to_act(xq.to(dt).to(DEV), dt), wu, bu, wc, bc, g32, be32, rm, rv, comp, ACT_RELU, dt)
    close(a, a_ref.detach(), 3e-2 if bf else 2e-4, "activation")
    dx, dwu, dbu, dwc, dbc, dg, dbe = ops.upconv_luconv_backward(sv, ops.to_act(daq.to(dt).to(DEV), dt), wu, bu, wc, bc, g32, comp, dt)
    # bf16: the ReLU mask is taken from the bf16 y0; ~0.4 % of the elements sit within one rounding step of bn(y0) = 0 and flip their
    # mask w.r.t. the float64 run, each flip moves a gradient term by O(1) -> ~sqrt(16)/sqrt(4096) = 6 % of a 4096-term sum (measured
    # 8 %).  The float32 run (no flips) pins the algebra to 5e-4.
    tol = 1.5e-1 if bf else 5e-4
    close(dx, xr.grad, tol, "dx")
    close(dwu, pr[0].grad, tol, "dw_up")
    close(dwc, pr[2].grad, tol, "dw0")
    close(dbu, pr[1].grad, tol, "db_up")
    close(dg, pr[4].grad, tol, "dgamma")
    close(dbe, pr[5].grad, tol, "dbeta")
    assert float(dbc.abs().max()) == 0.0


@pytest.mark.parametrize("geom", [(32, 32, 32, 16, 128, 128, 64), (32, 16, 16, 8, 256, 256, 128), (32, 8, 8, 4, 512, 512, 256)])
def test_full_size_composed_operator_adjoint_identities_bf16(geom):
    """BASELINE C2 sizes (b = 32: up_tr64 / up_tr128 / up_tr256 of a global view), bf16 -- no CPU reference is affordable there, but the
    composed operator's four kernels must be mutually adjoint.  y = A(w_up, w0) x + B(b_up, w0) + b0 is linear in x, in w0 and in
    (w_up, b_up) jointly, so with dy arbitrary
        <y - y|x=0, dy> == <x, dgrad(dy)>,   <y - b0, dy> == <w0, dw0> == <w_up, dw_up> + <b_up, db_up>
    up to the bf16 rounding of y, dx and of the composed weights (a size-independent property; the wide-brick instantiations -- along
    (D, H, W) and (D, W, H) -- and the gather kernels each serve one of the three shapes)."""
    N, D, H, W, Ci, Cm, Co = geom
    dt = torch.bfloat16
    L, s = lib(), stream_handle()
    g = torch.Generator(device=DEV).manual_seed(3)
    x = ops.new_act(N, D, H, W, Ci, dt, DEV)
    dy = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, DEV)
    x.normal_(generator=g)
    dy.normal_(generator=g)
    w_up = torch.randn(Ci, Cm, 2, 2, 2, device=DEV, generator=g) * 0.05
    b_up = torch.randn(Cm, device=DEV, generator=g) * 0.5
    w0 = torch.randn(Co, Cm, 3, 3, 3, device=DEV, generator=g) * 0.03
    b0 = torch.zeros(Co, device=DEV)
    comp = ops.ComposedUpConv()
    wf, wd, tab = comp.get(w_up, b_up, w0, b0, dt, geom=(N, D, H, W))

    def fwd(xin):
        y = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, DEV)
        L.call("pcrl_upconv_fwd", xin, wf, comp.w3f, tab, y, None, N, D, H, W, Ci, Co, dtype_code(dt), s)
        return y
    y = fwd(x)
    y_bias = fwd(torch.zeros_like(x))                      # B(b_up, w0): the bias table over the border classes
    # dy correlated with y (cosine ~0.9): with an independent random dy the inner products are ~1e-4 of |y||dy| and the identities would hold
    # to the tolerance below whatever the kernels did
    dy = ((y.float() / y.float().std()) + 0.5 * dy.float()).to(dt)
    dx = ops.new_act(N, D, H, W, Ci, dt, DEV)
    L.call("pcrl_upconv_dgrad", dy, wd, comp.wd3, dx, N, D, H, W, Ci, Co, dtype_code(dt), s)
    dwu, dbu, dw0 = torch.empty_like(w_up), torch.empty_like(b_up), torch.empty_like(w0)
    nb = L.call("pcrl_upconv_wgrad_ws_bytes", N, D, H, W, Ci, Cm, Co, dtype_code(dt))
    L.call("pcrl_upconv_wgrad", x, dy, w_up, b_up, w0, dwu, dbu, dw0, ops.workspace(nb, torch.device(DEV)), nb, N, D, H, W, Ci, Cm, Co, dtype_code(dt), s)
    dyd = dy.double()
    a_lin = float(((y.double() - y_bias.double()) * dyd).sum())
    a_all = float((y.double() * dyd).sum())
    b_x = float((x.double() * dx.double()).sum())
    c_w0 = float((w0.double() * dw0.double()).sum())
    c_up = float((w_up.double() * dwu.double()).sum() + (b_up.double() * dbu.double()).sum())
    scale = float(y.double().norm() * dyd.norm())
    print(f"  composed adjoint {geom}: <Ax,dy>={a_lin:.6e} <x,dx>={b_x:.6e} | <y,dy>={a_all:.6e} <w0,dw0>={c_w0:.6e} <w_up,dw_up>+<b_up,db_up>={c_up:.6e} "
          f"(|y||dy|={scale:.3e}: {abs(a_lin - b_x) / scale:.1e} {abs(a_all - c_w0) / scale:.1e} {abs(a_all - c_up) / scale:.1e})")
    assert abs(a_all) > 0.5 * scale
    assert abs(a_lin - b_x) < 5e-4 * scale
    assert abs(a_all - c_w0) < 2e-3 * scale and abs(a_all - c_up) < 2e-3 * scale


@pytest.mark.parametrize("geom", [(2, 4, 8, 16, 64, 64, 64), (3, 8, 16, 32, 128, 128, 64), (2, 4, 16, 8, 64, 64, 64), (1, 8, 32, 8, 128, 64, 128),
                                  (2, 4, 32, 24, 64, 64, 32), (1, 4, 8, 16, 64, 64, 256),
                                  # the 4 x 8 x 8-brick instantiations: natural and permuted axes, 64- and 32-channel output tiles of the data gradient
                                  (2, 8, 8, 8, 128, 128, 64), (3, 8, 8, 4, 64, 64, 64), (2, 16, 8, 4, 64, 64, 128), (25, 8, 8, 8, 256, 32, 32),
                                  (2, 8, 8, 4, 512, 64, 256)])
def test_brick_instantiations_equal_the_gather_kernels_to_an_ulp(geom):
    """bf16: the composed operator's wide-brick instantiations (forward <64, 1>, data gradient <64, 2>, along (D, H, W) and (D, W, H)) and the
    brick weight-gradient kernel against the GATHER kernels on the same operands (`pcrl_debug_set_conv_impl(1)` / `_wgrad_impl(1)`).  Both
    sides add the same bf16 products in float32, in another order: after the output rounding at most a fraction of a percent of the elements
    differ, by one bf16 ulp; the float32 results (statistics rows, gradient of the composed weights) agree to 1e-5.  The gather kernels are the
    ones the exact-float32 mode pins to 1e-4 against float64 above -- a wrong tap, phase, parity or border class in a brick instantiation
    would be an O(1) difference here, which the 2e-2 bf16 tolerance of the float64 comparison could hide on a thin border."""
    N, D, H, W, Ci, Cm, Co = geom
    dt = torch.bfloat16
    L, s = lib(), stream_handle()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = ops.new_act(N, D, H, W, Ci, dt, DEV)
    dy = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, DEV)
    x.normal_(generator=g)
    dy.normal_(generator=g)
    w_up = torch.randn(Ci, Cm, 2, 2, 2, device=DEV, generator=g) * 0.05
    b_up = torch.randn(Cm, device=DEV, generator=g) * 0.5
    w0 = torch.randn(Co, Cm, 3, 3, 3, device=DEV, generator=g) * 0.03
    b0 = torch.randn(Co, device=DEV, generator=g) * 0.1
    comp = ops.ComposedUpConv()
    wf, wd, tab = comp.get(w_up, b_up, w0, b0, dt, geom=(N, D, H, W))
    outs, used = [], []
    try:
        for impl in (0, 1):
            L.debug_set_conv_impl(impl)
            L.debug_set_wgrad_impl(impl)
            code = dtype_code(dt)
            used.append((L.call("pcrl_upconv_fwd_uses_brick", N, D, H, W, Ci, Co, code), L.call("pcrl_upconv_dgrad_uses_brick", N, D, H, W, Ci, Co, code),
                         L.call("pcrl_upconv_wgrad_uses_brick", N, D, H, W, Ci, Co, code)))
            rows = L.call("pcrl_upconv_stats_rows", N, D, H, W, Ci, Co, code)
            y = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, dt, DEV)
            part = torch.zeros(rows, Co, 2, device=DEV)
            L.call("pcrl_upconv_fwd", x, wf, comp.w3f, tab, y, part, N, D, H, W, Ci, Co, code, s)
            dx = ops.new_act(N, D, H, W, Ci, dt, DEV)
            L.call("pcrl_upconv_dgrad", dy, wd, comp.wd3, dx, N, D, H, W, Ci, Co, code, s)
            dweff, box = torch.empty(64 * Ci * Co, device=DEV), torch.empty(27 * Co, device=DEV)
            nb = L.call("pcrl_upconv_wgrad_accum_ws_bytes", N, D, H, W, Ci, Co, code)
            L.call("pcrl_upconv_wgrad_accum", x, dy, dweff, box, 1, ops.workspace(nb, torch.device(DEV)), nb, N, D, H, W, Ci, Co, code, s)
            outs.append((y.float(), dx.float(), part.double().sum(0), dweff.double(), box.double()))
    finally:
        L.debug_set_conv_impl(0)
        L.debug_set_wgrad_impl(0)
    assert used[1] == (0, 0, 0) and any(used[0]), used      # every shape here runs at least one brick kernel by default
    (ya, dxa, pa, wa, ba), (yb, dxb, pb, wb, bb) = outs
    for name, a, b in (("y0", ya, yb), ("dx", dxa, dxb)):
        d = (a - b).abs()
        frac = float((d > 0).float().mean())
        ulp = float((d / b.abs().clamp_min(1e-2)).max())
        assert frac < 5e-3 and ulp <= 2.0 ** -7 + 1e-6, (name, used, frac, ulp)
    assert float(((pa - pb).abs() / pb.abs().clamp_min(1e-3)).max()) < 1e-5
    assert float((wa - wb).abs().max()) < 1e-5 * float(wb.abs().max())
    assert float((ba - bb).abs().max()) < 1e-5 * max(float(bb.abs().max()), 1e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(2, 3, 4, 5, 32, 32), (3, 1, 2, 3, 32, 64), (1, 4, 8, 16, 64, 64), (2, 8, 8, 4, 64, 32)])
def test_border_class_sums_from_the_border_when_dy_sums_to_zero(geom, dt):
    """pcrl_upconv_wgrad_accum flags bit 1: dy0 sums to zero per channel (the output of a training-mode BatchNorm backward), so the class sums
    read only the border voxels and the interior class is minus the rest.  On a dy0 made zero-sum in float64 before the rounding to the
    activation dtype, both routes give the same box sums up to the rounding of dy0 (float32: 1e-5 of the largest; bfloat16: the full sum
    carries the rounding noise of every voxel, the border route does not -- they agree to 2^-9 * sqrt(#voxels) * rms); the gradient of the
    composed weights is untouched by the flag."""
    N, D, H, W, Ci, Co = geom
    L, s = lib(), stream_handle()
    x = ops.to_act(rnd(N, Ci, D, H, W, seed=1).to(dt).to(DEV), dt)
    dy = rnd(N, Co, 2 * D, 2 * H, 2 * W, seed=2)
    dy = dy - dy.mean(dim=(0, 2, 3, 4), keepdim=True)
    dya = ops.to_act(dy.to(dt).to(DEV), dt)
    outs = []
    for flags in (1, 3):
        dweff, box = torch.empty(64 * Ci * Co, device=DEV), torch.empty(27 * Co, device=DEV)
        nb = L.call("pcrl_upconv_wgrad_accum_ws_bytes", N, D, H, W, Ci, Co, dtype_code(dt))
        L.call("pcrl_upconv_wgrad_accum", x, dya, dweff, box, flags, ops.workspace(nb, torch.device(DEV)), nb, N, D, H, W, Ci, Co, dtype_code(dt), s)
        outs.append((back(dweff), back(box)))
    (wa, ba), (wb, bb) = outs
    assert torch.equal(wa, wb)
    M = N * 8 * D * H * W
    noise = (2.0 ** -9 if dt == torch.bfloat16 else 2.0 ** -24) * (M ** 0.5) * float(dy.std()) * 4 + 1e-5 * float(ba.abs().max())
    assert float((ba - bb).abs().max()) <= noise, (float((ba - bb).abs().max()), noise)
    # and against float64: box[t] = sum of dy over the fine voxels from which tap t stays inside the grid
    dq = dy.to(dt).double()
    FD, FH, FW = 2 * D, 2 * H, 2 * W
    ref = torch.zeros(27, Co, dtype=torch.float64)
    for t in range(27):
        td, th, tw = t // 9 - 1, (t // 3) % 3 - 1, t % 3 - 1
        sl = lambda k, n_: slice(max(0, -k), n_ - max(0, k))
        ref[t] = dq[:, :, sl(td, FD), sl(th, FH), sl(tw, FW)].sum(dim=(0, 2, 3, 4))
    for got, what in ((ba, "all voxels"), (bb, "border voxels")):
        assert float((got.view(27, Co) - ref).abs().max()) <= noise, what


def test_tiled_weight_pack_is_bit_identical_to_the_untiled_one(tmp_path):
    """pcrl_upconv_compose writes the composed weights in four layouts; for Ci % 32 == Co % 32 == 0 the pack runs tiled (the two K-contiguous forms
    through an LDS transposition, upc_pack_tiled_kernel).  PCRL_UPC_PACK_TILED=0 selects the untiled kernel (read once per process): two
    subprocesses, the same closed-form weights, all four forms and the bias table byte for byte -- bf16 and float32."""
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    prog = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from pcrlv2_amd._lib import dtype_code, lib, stream_handle
L = lib()
out = []
for dt in (torch.bfloat16, torch.float32):
    for Ci, Cm, Co in ((64, 64, 32), (128, 128, 64), (32, 96, 160)):
        g = torch.Generator().manual_seed(Ci + Co)
        w_up = (torch.rand(Ci, Cm, 2, 2, 2, generator=g) - 0.5).cuda(); b_up = (torch.rand(Cm, generator=g) - 0.5).cuda()
        w0 = (torch.rand(Co, Cm, 3, 3, 3, generator=g) - 0.5).cuda(); b0 = (torch.rand(Co, generator=g) - 0.5).cuda()
        wf = torch.empty(64 * Ci * Co, dtype=dt, device="cuda"); wd = torch.empty_like(wf)
        w3 = torch.empty(216 * Ci * Co, dtype=dt, device="cuda"); wd3 = torch.empty_like(w3)
        tab = torch.empty(27 * Co, device="cuda")
        nb = L.call("pcrl_upconv_compose_ws_bytes", Ci, Cm, Co, dtype_code(dt))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        L.call("pcrl_upconv_compose", w_up, b_up, w0, b0, wf, wd, w3, wd3, tab, ws, nb, Ci, Cm, Co, dtype_code(dt), stream_handle())
        torch.cuda.synchronize()
        for t in (wf, wd, w3, wd3, tab):
            out.append(hashlib.sha256(t.view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16])
print("HASHES", " ".join(out))
''' % root
    res = []
    for flag in ("1", "0"):
        env = dict(os.environ, PCRL_UPC_PACK_TILED=flag)
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASHES")][0])
    assert res[0] == res[1], (res[0], res[1])
    assert len(res[0].split()) == 1 + 2 * 3 * 5
