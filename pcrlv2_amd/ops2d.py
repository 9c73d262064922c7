"""Host-side wrappers of the 2D (PCRLv2 ResNet-18 U-Net) entry points of libpcrl_hip.so  --  SURVEY 8f N1.

Activations are logical [N, C, H, W] tensors in NHWC memory (torch.channels_last), float32 or bfloat16.  Everything here
launches kernels of the C ABI (include/pcrl_hip.h, "2D path"); there is no eager / CPU fallback.
"""
from __future__ import annotations

import torch

from . import config, ops
from ._lib import ACT_RELU, PcrlError, dtype_code, lib, stream_handle


def new_act2(N, H, W, C, dtype, device) -> torch.Tensor:
    return torch.empty((N, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def dims2(t: torch.Tensor):
    """(N, H, W, C) of an NHWC-memory activation."""
    if t.dim() != 4:
        raise PcrlError(f"expected a 4-D activation, got shape {tuple(t.shape)}")
    N, C, H, W = t.shape
    if not t.permute(0, 2, 3, 1).is_contiguous():
        raise PcrlError("activation is not NHWC (channels_last) contiguous")
    return N, H, W, C


def to_act2(x: torch.Tensor, dtype, pad_to: int = 0) -> torch.Tensor:
    """API-boundary glue: any [N,C,H,W] tensor -> NHWC memory in `dtype`, channels zero-padded to `pad_to` (3-channel images and
    3-channel gradients feed kernels that want a power-of-two channel count >= 8)."""
    N, C, H, W = x.shape
    if pad_to and C < pad_to:
        out = torch.zeros((N, H, W, pad_to), dtype=dtype, device=x.device)
        out[..., :C] = x.permute(0, 2, 3, 1)
        return out.permute(0, 3, 1, 2)
    if x.dtype != dtype:
        x = x.to(dtype)
    if not x.permute(0, 2, 3, 1).is_contiguous():
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return x


def _pow2_at_least_8(c: int) -> int:
    p = 8
    while p < c:
        p *= 2
    return p


class PackedConv2d:
    """K-contiguous copies (forward and data-gradient orientation) of one nn.Conv2d weight in the activation dtype."""

    def __init__(self):
        self.key = None
        self.fwd = None
        self.dgrad = None
        self.s2_key = None
        self.s2 = None        # stride-2 data gradient: {(a, b): parity-class pack}
        self._guard, self._guard_s2 = ops._CacheGuard(), ops._CacheGuard()   # built on one stream, read on another (config.VIEW_STREAMS_2D)

    def get_s2(self, w: torch.Tensor, dtype):
        """Parity-class packs of the stride-2 data gradient (pcrl_conv2d_dgrad_s2): 3x3 -> four classes, 1x1 -> class (0, 0)."""
        key = (ops._weights_epoch, w._version, w.data_ptr(), dtype)
        if key != self.s2_key:
            L, s = lib(), stream_handle()
            Co, Ci, KH, KW = w.shape
            CoP = _pow2_at_least_8(Co)
            self.s2 = {}
            for a, b in (((0, 0), (0, 1), (1, 0), (1, 1)) if KH == 3 else ((0, 0),)):
                taps = ((2 if a else 1) * (2 if b else 1)) if KH == 3 else 1
                t = torch.empty(L.call("pcrl_conv2d_packed_elems", Ci, taps, CoP), dtype=dtype, device=w.device)
                L.call("pcrl_conv2d_pack_s2", w.detach(), t, Co, Ci, KH, KW, CoP, a, b, dtype_code(dtype), s)
                self.s2[(a, b)] = t
            self.s2_key = key
            self._guard_s2._built(w.device)
        else:
            self._guard_s2._reading(w.device)
        return self.s2

    def get(self, w: torch.Tensor, dtype, CiP: int):
        key = (ops._weights_epoch, w._version, w.data_ptr(), dtype, CiP)
        if key != self.key:
            L, s = lib(), stream_handle()
            Co, Ci, KH, KW = w.shape
            CoP = _pow2_at_least_8(Co)
            self.fwd = torch.empty(L.call("pcrl_conv2d_packed_elems", Co, KH * KW, CiP), dtype=dtype, device=w.device)
            self.dgrad = torch.empty(L.call("pcrl_conv2d_packed_elems", Ci, KH * KW, CoP), dtype=dtype, device=w.device)
            L.call("pcrl_conv2d_pack", w.detach(), self.fwd, Co, Ci, KH, KW, CiP, 0, dtype_code(dtype), s)
            L.call("pcrl_conv2d_pack", w.detach(), self.dgrad, Co, Ci, KH, KW, CoP, 1, dtype_code(dtype), s)
            self.key = key
            self._guard._built(w.device)
        else:
            self._guard._reading(w.device)
        return self.fwd, self.dgrad


class PackedStem:
    """bf16 [64][7][32] pack of the ResNet stem's 7x7 weight for csrc/stem2d.hip (k = kw * 4 + c), cached per parameter version."""

    def __init__(self):
        self.key, self.w = None, None
        self._guard = ops._CacheGuard()

    def get(self, w: torch.Tensor):
        key = (ops._weights_epoch, w._version, w.data_ptr())
        if key != self.key:
            L = lib()
            self.w = torch.empty(L.call("pcrl_stem7_packed_elems"), dtype=torch.bfloat16, device=w.device)
            L.call("pcrl_stem7_pack", w.detach(), self.w, stream_handle())
            self.key = key
            self._guard._built(w.device)
        else:
            self._guard._reading(w.device)
        return self.w


def stem_ok(x, w, dtype) -> bool:
    """The dedicated stem kernels take this image / weight (float32 NCHW [N,3,H,W] on the GPU, 64 x 3 x 7 x 7, bf16 compute, H/2 % 8 == W/2 % 32 == 0)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous() and tuple(w.shape) == (64, 3, 7, 7)):
        return False
    return bool(lib().call("pcrl_stem7_ok", x.shape[0], x.shape[2], x.shape[3], dtype_code(dtype)))


def stem_forward(x, w, packed: PackedStem, dtype):
    """conv1 of the ResNet stem on the float32 NCHW image.  -> (y bf16 NHWC [N,64,H/2,W/2], stats_partial, rows)"""
    L = lib()
    N, _, H, W = x.shape
    y = new_act2(N, H // 2, W // 2, 64, dtype, x.device)
    rows = L.call("pcrl_stem7_stats_rows", N, H, W)
    partial = ops._f32(rows * 64 * 2, x.device)
    L.call("pcrl_stem7_fwd", x, packed.get(w), y, partial, N, H, W, dtype_code(dtype), stream_handle())
    return y, partial, rows


def stem_wgrad(x, dy, w, dtype):
    """float32 [64,3,7,7] weight gradient of the stem from the float32 NCHW image and dy (bf16 NHWC), on the weight-gradient side stream."""
    L = lib()
    N, _, H, W = x.shape
    dw = torch.empty_like(w, dtype=torch.float32, memory_format=torch.contiguous_format)
    nb = L.call("pcrl_stem7_wgrad_ws_bytes", N, H, W)
    with ops.side_wgrad(x.device, x, dy, path2d=True) as ws:
        L.call("pcrl_stem7_wgrad", x, dy, dw, ws(nb), nb, N, H, W, dtype_code(dtype), stream_handle())
    return dw


def out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def conv2d_forward(x, w, bias, packed: PackedConv2d, stride, pad, up, dtype, want_stats=True, out_f32=False, stats_only=False):
    """x: NHWC activation with CiP >= w.shape[1] channels.  -> (y, stats_partial | None, rows).  stats_only: the statistics rows are all the caller
    wants (a deep-supervision head's convolution whose map nothing reads): where the library can (pcrl_conv2d_fwd_stats_only_ok) y is not written and
    None is returned in its place."""
    L, s = lib(), stream_handle()
    N, Hi, Wi, CiP = dims2(x)
    Co, Ci, KH, KW = w.shape
    if CiP < Ci or x.dtype != dtype:
        raise PcrlError(f"conv2d_forward: input has {CiP} channels in {x.dtype}, weight wants {Ci} in {dtype}")
    wf, _ = packed.get(w, dtype, CiP)
    Hl, Wl = (2 * Hi, 2 * Wi) if up else (Hi, Wi)
    Ho, Wo = out_size(Hl, KH, stride, pad), out_size(Wl, KW, stride, pad)
    if stats_only and want_stats and L.call("pcrl_conv2d_fwd_stats_only_ok", N, Hi, Wi, CiP, Co, KH, KW, stride, pad, int(up), int(out_f32), dtype_code(dtype)):
        y = None
    else:
        y = new_act2(N, Ho, Wo, Co, torch.float32 if out_f32 else dtype, x.device)
    rows = L.call("pcrl_conv2d_fwd_stats_rows", N, Hi, Wi, CiP, Co, KH, KW, stride, pad, int(up), int(out_f32), dtype_code(dtype)) if want_stats else 0
    partial = ops._f32(rows * Co * 2, x.device) if want_stats else None
    L.call("pcrl_conv2d_fwd", x, wf, None if bias is None else bias.detach(), y, partial, rows, N, Hi, Wi, CiP, Co, KH, KW, stride, pad,
           int(up), int(out_f32), dtype_code(dtype), s)
    return y, partial, rows


def conv2d_backward(x, dy, w, packed: PackedConv2d, stride, pad, up, dtype, need_dx=True, below=None):
    """x: the forward input (NHWC, CiP channels); dy: NHWC gradient of the forward output in `dtype` with CoP = pow2 >= 8 channels
    (zero-padded by the caller when Co = 3).  -> (dx | None, dw float32 [Co][Ci][KH][KW])
    `below` = {"y": pre-normalisation tensor, "c": (mean, rstd, scale, shift)} of the BatchNorm + ReLU layer whose activation x is, when this
    convolution is that activation's ONLY consumer (conv2 over relu(bn1(conv1)) in a BasicBlock / DecoderBlock): where the library has the kernel
    (pcrl_conv2d_dgrad_bnred_rows) the data gradient takes the first pass of that layer's BatchNorm backward from its output tiles and leaves
    below["pre"] = (partial, rows) for ops.bn_act_backward(pre_partial=...)."""
    L, s = lib(), stream_handle()
    N, Hi, Wi, CiP = dims2(x)
    _, Ho, Wo, CoP = dims2(dy)
    Co, Ci, KH, KW = w.shape
    nb = L.call("pcrl_conv2d_wgrad_ws_bytes", N, Ho, Wo, CiP, CoP, KH, KW)
    dw_full = torch.empty((CoP, Ci, KH, KW), dtype=torch.float32, device=x.device)
    with ops.side_wgrad(x.device, x, dy, path2d=True) as ws:      # weight gradients run next to the data-gradient / BatchNorm chain (ops.side_wgrad)
        L.call("pcrl_conv2d_wgrad", x, dy, dw_full, ws(nb), nb, N, Hi, Wi, CiP, Ci, Ho, Wo, CoP, KH, KW, stride, pad,
               int(up), dtype_code(dtype), stream_handle())
    dw = dw_full if CoP == Co else dw_full[:Co]
    dx = None
    if need_dx and stride == 2 and not up and Hi % 2 == 0 and Wi % 2 == 0 and ((KH == 3 and pad == 1) or (KH == 1 and pad == 0)) and KH == KW \
            and _pow2_at_least_8(Co) == CoP:
        # stride 2: four parity classes of dx, each a small stride-1 gather over dy (no idle taps); 1x1: three classes are zero
        packs = packed.get_s2(w, dtype)
        dx = new_act2(N, Hi, Wi, Ci, dtype, x.device)
        if KH == 1:
            L.call("pcrl_zero", dx, dx.numel() * dx.element_size(), s)      # 1x1 stride 2: three of the four parity classes receive nothing
        for (a, b), wp in packs.items():
            L.call("pcrl_conv2d_dgrad_s2", dy, wp, dx, N, Hi, Wi, Ci, Ho, Wo, CoP, KH, KW, a, b, dtype_code(dtype), s)
    elif need_dx:
        _, wd = packed.get(w, dtype, CiP)
        if up and KH == 3 and stride == 1 and pad == 1 and L.call("pcrl_conv2d_dgrad_up_ok", N, Hi, Wi, Ci, CoP, dtype_code(dtype)):
            # the narrow layers: data gradient and the nearest x2 upsample's backward in one kernel -- the fine-resolution gradient is never stored
            dx = new_act2(N, Hi, Wi, Ci, dtype, x.device)
            L.call("pcrl_conv2d_dgrad_up", dy, wd, dx, N, Hi, Wi, Ci, CoP, dtype_code(dtype), s)
            return dx, dw
        Hl, Wl = (2 * Hi, 2 * Wi) if up else (Hi, Wi)
        dxl = new_act2(N, Hl, Wl, Ci, dtype, x.device)
        brows = 0
        if (below is not None and config.DGRAD_BNRED and not up and KH == 3 and KW == 3 and stride == 1 and pad == 1 and (Ho, Wo) == (Hi, Wi)
                and below["y"].dtype == dtype and tuple(below["y"].shape) == tuple(dxl.shape)):
            brows = L.call("pcrl_conv2d_dgrad_bnred_rows", N, Hi, Wi, Ci, CoP, ACT_RELU, dtype_code(dtype))
        if brows:
            mean, rstd, scale, shift = below["c"]
            bpart = ops._f32(brows * Ci * 2, x.device)
            L.call("pcrl_conv2d_dgrad_bnred", dy, wd, dxl, below["y"], scale, shift, mean, rstd, bpart, N, Hi, Wi, Ci, CoP, ACT_RELU, dtype_code(dtype), s)
            below["pre"] = (bpart, brows)
            return dxl, dw
        L.call("pcrl_conv2d_dgrad", dy, wd, dxl, N, Hl, Wl, Ci, Ho, Wo, CoP, KH, KW, stride, pad, dtype_code(dtype), s)
        if up:
            dx = new_act2(N, Hi, Wi, Ci, dtype, x.device)
            L.call("pcrl_upsample2d_nearest2_bwd", dxl, dx, N, Hi, Wi, Ci, dtype_code(dtype), s)
        else:
            dx = dxl
    return dx, dw


def colsum(v, M, C, dtype):
    """float32 [C] column sums of an [M][C] tensor (bias gradients)."""
    L = lib()
    out = ops._f32(C, v.device)
    nb = L.call("pcrl_colsum_ws_bytes", M, C)
    L.call("pcrl_colsum", v, out, ops.workspace(nb, v.device), nb, M, C, dtype_code(dtype), stream_handle())
    return out


def maxpool_forward(x, dtype):
    N, H, W, C = dims2(x)
    Ho, Wo = out_size(H, 3, 2, 1), out_size(W, 3, 2, 1)
    y = new_act2(N, Ho, Wo, C, dtype, x.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    lib().call("pcrl_maxpool2d_3s2_fwd", x, y, idx, N, H, W, C, dtype_code(dtype), stream_handle())
    return y, idx


def maxpool_backward(dy, idx, in_dims, dtype):
    N, H, W, C = in_dims
    dx = new_act2(N, H, W, C, dtype, dy.device)
    lib().call("pcrl_maxpool2d_3s2_bwd", dy, idx, dx, N, H, W, C, dtype_code(dtype), stream_handle())
    return dx


def bilinear_forward(x, scale: int):
    """x: float32 NHWC-memory [N,C,H,W] map."""
    N, H, W, C = dims2(x)
    y = new_act2(N, H * scale, W * scale, C, torch.float32, x.device)
    lib().call("pcrl_upsample2d_bilinear_fwd", x, y, N, H, W, C, scale, stream_handle())
    return y


def bilinear_backward(dy, in_dims, scale: int):
    N, H, W, C = in_dims
    dx = new_act2(N, H, W, C, torch.float32, dy.device)
    lib().call("pcrl_upsample2d_bilinear_bwd", dy, dx, N, H, W, C, scale, stream_handle())
    return dx


def add_relu_forward(t, r, dtype):
    a = torch.empty_like(t)
    lib().call("pcrl_add_relu_fwd", t, r, a, t.numel(), dtype_code(dtype), stream_handle())
    return a


def relu_mask_backward(da, a, dtype):
    g = torch.empty_like(a)
    lib().call("pcrl_relu_mask_bwd", da, a, g, a.numel(), dtype_code(dtype), stream_handle())
    return g


def gap_forward(a, dtype):
    N, H, W, C = dims2(a)
    g = ops._f32(N * C, a.device).view(N, C)
    L = lib()
    nb = L.call("pcrl_gap_ws_bytes", N, H * W, C)
    L.call("pcrl_gap_fwd", a, g, ops.workspace(nb, a.device), nb, N, H * W, C, dtype_code(dtype), stream_handle())
    return g


def gap_backward(dg, like, dtype, add_src=None):
    """da = add_src + dg / (H*W) broadcast over the pixels (add_src may be None)."""
    N, H, W, C = dims2(like)
    da = torch.empty_like(like)
    lib().call("pcrl_gap_bwd", dg.contiguous(), add_src, da, N, H * W, C, dtype_code(dtype), stream_handle())
    return da


# ----------------------------------------------------------------------------------------------
# the 3-channel ends of the step (csrc/heads2d.hip)
# ----------------------------------------------------------------------------------------------
def _nhwc_f32(p):
    """An [N,C,H,W] float32 tensor in NHWC memory (what the segmentation / deep-supervision heads write), as is or converted."""
    return to_act2(p, torch.float32)


def _nchw_f32(gt, device):
    gt = gt.to(device=device, dtype=torch.float32)
    return gt if gt.is_contiguous() else gt.contiguous()


def mse2d_forward(p, gt):
    """nn.MSELoss()(p, gt): p float32 [N,C,H,W] in NHWC memory, gt float32 [N,C,H,W] contiguous (the loader's layout).  -> 0-d loss"""
    L = lib()
    N, H, W, C = dims2(p)
    loss = ops._f32(1, p.device)
    nb = L.call("pcrl_mse2d_ws_bytes", N * H * W)
    L.call("pcrl_mse2d_fwd", p, gt, loss, ops.workspace(nb, p.device), nb, N, H * W, C, stream_handle())
    return loss.view(())


def mse2d_backward(p, gt, dloss, CP, dtype):
    """d loss / d p as `dtype` [N,CP,H,W] in NHWC memory (channels C..CP-1 zero) + its per-channel sums float32 [C] (the bias gradient of the
    convolution that produced p)."""
    L = lib()
    N, H, W, C = dims2(p)
    M = N * H * W
    dy = new_act2(N, H, W, CP, dtype, p.device)
    rows = L.call("pcrl_rows1024", M)
    colpart = ops._f32(rows * CP, p.device)
    L.call("pcrl_mse2d_bwd_pad", p, gt, dloss.detach().reshape(1).float(), dy, colpart, N, H * W, C, CP, dtype_code(dtype), stream_handle())
    return dy, colpart, rows


def colsum_f32(partial, rows, C):
    """float32 [C] column sums of a float32 [rows][C] matrix (second stage of the block partials above)."""
    L = lib()
    out = ops._f32(C, partial.device)
    nb = L.call("pcrl_colsum_ws_bytes", rows, C)
    L.call("pcrl_colsum", partial, out, ops.workspace(nb, partial.device), nb, rows, C, dtype_code(torch.float32), stream_handle())
    return out


def conv1x1_small_backward(x, dy, w, dtype):
    """Backward of nn.Conv2d(Ci, 3, 1) (deep_supervision_head[3], pcrlv2_model.py:106) in one pass: x `dtype` NHWC [N,Ci,H,W], dy float32 NHWC
    [N,3,H,W], w float32 [3,Ci,1,1] -> (dx like x, dw float32 [3,Ci,1,1], db float32 [3])."""
    L = lib()
    N, H, W, Ci = dims2(x)
    M = N * H * W
    Co = w.shape[0]
    rows = L.call("pcrl_rows1024", M)
    width = Co * Ci + Co
    PW = 4
    while PW < width:       # pcrl_colsum takes float32 rows of 4 * 2^k columns
        PW *= 2
    part = ops._f32(rows * PW, x.device)
    dx = torch.empty_like(x)
    L.call("pcrl_conv2d_1x1_small_bwd", x, dy, w.detach(), dx, part, M, Ci, Co, PW, dtype_code(dtype), stream_handle())
    tot = colsum_f32(part, rows, PW)
    return dx, tot[:Co * Ci].view(Co, Ci, 1, 1), tot[Co * Ci:width]


def image_to_act(x, dtype, CP=8):
    """The network input: float32 NCHW [N,3,H,W] -> `dtype` NHWC [N,CP,H,W] with zero padding channels, one launch (the stem's gather wants a
    power-of-two channel count; was torch.zeros + a strided copy)."""
    N, C, H, W = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C <= CP):
        return to_act2(x, dtype, pad_to=CP)
    out = new_act2(N, H, W, CP, dtype, x.device)
    lib().call("pcrl_nchw_to_nhwc_pad", x, out, N, C, H * W, CP, dtype_code(dtype), stream_handle())
    return out
