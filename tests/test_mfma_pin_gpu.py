"""Tight pin of the bf16 MFMA kernels to the REFERENCE (the throughput kernels were held to the reference only through
bf16-class tolerances; the float32-class pin ran on the gather kernel).

tests/golden/mfma_pin.npz (oracle/make_golden.py --mfma-pin) holds what the real `LUConv` / `UpTransition` modules of
models/pcrlv2_model_3d.py compute in float64 on operands that are EXACTLY representable in bfloat16 (pcrlv2_oracle.mfma_pin_*_case; for the
composed operator the weights are chosen so that the composed 8-tap phase weights are bf16-exact too).  A bf16 MFMA kernel then forms exact
products and differs from those numbers by float32 accumulation order alone, so:
  * float32 outputs -- the (sum, sum^2) BatchNorm rows against the running statistics the reference MODULE holds after one training-mode
    forward, the weight gradient against autograd of the module's conv1 -- are held at float32-class tolerance (1e-5 / 3e-5 relative);
  * bf16 outputs (convolution output, data gradient) are held to "the correctly rounded reference value": |got - ref| <= half a bf16 ulp of
    ref + 2e-5 * max|ref| (the accumulation-order slack) at every sampled element -- a wrong tap, border class or phase is off by ~1e-1.
Every case asserts that the kernel that ran is the wide-brick one (`pcrl_conv3d_k3_fwd_kernel` == 2, `pcrl_upconv_*_uses_brick` == 1)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

import pcrlv2_oracle as O  # noqa: E402
from make_golden import sample_idx  # noqa: E402
from pcrlv2_amd import ops  # noqa: E402
from pcrlv2_amd._lib import dtype_code, lib, stream_handle  # noqa: E402

DEV = torch.device("cuda")
BF = torch.bfloat16
FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "mfma_pin.npz"))
K = 4096


def samples(t, seed):
    f = t.detach().double().cpu().reshape(-1).numpy()
    return f[sample_idx(f.size, K, seed)]


def ncdhw(act):
    """NDHWC-memory activation (logical NCDHW) -> contiguous NCDHW float64 on the CPU (the layout the fixture's sample indices address)"""
    return act.detach().double().cpu().contiguous()


def assert_correctly_rounded(got, ref, vmax, what):
    """bf16 output vs float64 reference: within half a bf16 ulp of the reference value plus the float32 accumulation slack."""
    ref = np.asarray(ref, dtype=np.float64)
    ulp = np.exp2(np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - 7)
    tol = 0.5 * ulp + 2e-5 * vmax
    err = np.abs(got - ref)
    bad = err > tol
    exact = float(np.mean(got == torch.from_numpy(ref).to(BF).double().numpy()))
    assert not bad.any(), f"{what}: {int(bad.sum())} of {len(ref)} samples off by more than half an ulp + 2e-5 max (worst {float((err / tol).max()):.1f} x tol)"
    assert exact >= 0.99, f"{what}: only {exact:.4f} of the samples equal the correctly rounded reference"
    return exact


def assert_stats(part, M, name, what):
    """(sum, sum^2) rows of the float32 accumulators vs the running statistics the reference module holds after one forward
    (running_mean = 0.1 * mean, running_var = 0.9 + 0.1 * unbiased variance: nn.BatchNorm3d defaults, momentum 0.1)."""
    st = part.double().sum(0).cpu().numpy()
    mean = st[:, 0] / M
    var = st[:, 1] / M - mean * mean
    ref_mean = FX[name + ".running_mean"] / 0.1
    ref_var = (FX[name + ".running_var"] - 0.9) / 0.1 * (M - 1) / M
    e_m = np.abs(mean - ref_mean).max() / max(np.abs(ref_mean).max(), np.sqrt(ref_var.max()))
    e_v = (np.abs(var - ref_var) / ref_var).max()
    assert e_m < 1e-5 and e_v < 2e-5, (what, e_m, e_v)
    return e_m, e_v


@pytest.mark.parametrize("name", list(O.MFMA_PIN_CONV))
def test_wide_brick_conv_kernels_against_the_reference_luconv(name):
    N, (D, H, W), Ci, Co = O.MFMA_PIN_CONV[name]
    L, s = lib(), stream_handle()
    c = O.mfma_pin_conv_case(name)
    M = N * D * H * W
    assert L.call("pcrl_conv3d_k3_fwd_kernel", N, D, H, W, Ci, Co, dtype_code(BF)) == 2, "forward is expected on the wide-brick kernel"
    assert L.call("pcrl_conv3d_k3_fwd_kernel", N, D, H, W, Co, Ci, dtype_code(BF)) == 2, "data gradient is expected on the wide-brick kernel"
    xa = ops.to_act(c["x"].to(BF).to(DEV), BF)
    dya = ops.to_act(c["dy"].to(BF).to(DEV), BF)
    assert torch.equal(ncdhw(xa), c["x"])                                  # the operands reach the kernel unrounded
    wf, wd = ops.PackedWeights("conv3").get(c["w"].float().to(DEV), BF)
    rows = L.call("pcrl_conv3d_k3_stats_rows", N, D, H, W, Ci, Co, dtype_code(BF))
    y = ops.new_act(N, D, H, W, Co, BF, DEV)
    part = torch.zeros(rows, Co, 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_k3_fwd", xa, wf, c["b"].float().to(DEV), y, part, N, D, H, W, Ci, Co, dtype_code(BF), s)
    ex_y = assert_correctly_rounded(samples(ncdhw(y), 3), FX[name + ".y"], float(FX[name + ".y_max"]), name + " y")
    e_m, e_v = assert_stats(part, M, name, name + " statistics")
    dx = ops.new_act(N, D, H, W, Ci, BF, DEV)
    L.call("pcrl_conv3d_k3_fwd", dya, wd, None, dx, None, N, D, H, W, Co, Ci, dtype_code(BF), s)
    ex_dx = assert_correctly_rounded(samples(ncdhw(dx), 4), FX[name + ".dx"], float(FX[name + ".dx_max"]), name + " dx")
    nb = L.call("pcrl_conv3d_k3_wgrad_ws_bytes", N, D, H, W, Ci, Co)
    dw = torch.zeros(Co, Ci, 3, 3, 3, dtype=torch.float32, device=DEV)
    L.call("pcrl_conv3d_k3_wgrad", xa, dya, dw, ops.workspace(nb, DEV), nb, N, D, H, W, Ci, Co, dtype_code(BF), s)
    e_dw = np.abs(samples(dw, 5) - FX[name + ".dw"]).max() / float(FX[name + ".dw_max"])
    e_l2 = abs(float(dw.double().norm()) - float(FX[name + ".dw_l2"])) / float(FX[name + ".dw_l2"])
    assert e_dw < 3e-5 and e_l2 < 1e-5, (name, "weight gradient", e_dw, e_l2)
    print(f"  {name}: y / dx equal to the correctly rounded reference at {ex_y:.4f} / {ex_dx:.4f} of {K} samples; mean {e_m:.1e} var {e_v:.1e} dW {e_dw:.1e} |dW| {e_l2:.1e}")


@pytest.mark.parametrize("name", list(O.MFMA_PIN_UP))
def test_composed_upconv_kernels_against_the_reference_uptransition(name):
    N, (D, H, W), C = O.MFMA_PIN_UP[name]
    Co = 64
    L, s = lib(), stream_handle()
    c = O.mfma_pin_up_case(name)
    assert L.call("pcrl_upconv_fwd_uses_brick", N, D, H, W, C, Co, dtype_code(BF)) == 1
    assert L.call("pcrl_upconv_dgrad_uses_brick", N, D, H, W, C, Co, dtype_code(BF)) == 1
    to_dev = lambda t: t.float().to(DEV).contiguous()
    comp = ops.ComposedUpConv()
    wf, wd, tab = comp.get(to_dev(c["w_up"]), to_dev(c["b_up"]), to_dev(c["w0"]), to_dev(c["b0"]), BF, geom=(N, D, H, W))
    xa = ops.to_act(c["x"].to(BF).to(DEV), BF)
    rows = L.call("pcrl_upconv_stats_rows", N, D, H, W, C, Co, dtype_code(BF))
    y = ops.new_act(N, 2 * D, 2 * H, 2 * W, Co, BF, DEV)
    part = torch.zeros(rows, Co, 2, dtype=torch.float32, device=DEV)
    L.call("pcrl_upconv_fwd", xa, wf, comp.w3f, tab, y, part, N, D, H, W, C, Co, dtype_code(BF), s)
    ex_y = assert_correctly_rounded(samples(ncdhw(y), 3), FX[name + ".y"], float(FX[name + ".y_max"]), name + " y0")
    e_m, e_v = assert_stats(part, 8 * N * D * H * W, name, name + " statistics")
    dya = ops.to_act(c["dy0"].to(BF).to(DEV), BF)
    dx = ops.new_act(N, D, H, W, C, BF, DEV)
    L.call("pcrl_upconv_dgrad", dya, wd, comp.wd3, dx, N, D, H, W, C, Co, dtype_code(BF), s)
    ex_dx = assert_correctly_rounded(samples(ncdhw(dx), 4), FX[name + ".dx"], float(FX[name + ".dx_max"]), name + " dx")
    print(f"  {name}: y0 / dx equal to the correctly rounded reference at {ex_y:.4f} / {ex_dx:.4f} of {K} samples; mean {e_m:.1e} var {e_v:.1e}")
