"""TEST INFRASTRUCTURE: float64 PyTorch restatement of the augmentation definitions that csrc/augment.hip implements
(torchio's documented behaviour of RandomFlip / RandomAffine / RandomBlur / RandomNoise / RandomGamma / RandomSwap / ZNormalization as
configured at data.py:73-89 of the reference).  Takes the SAME drawn parameters as pcrlv2_amd.data.apply_*; runs on any device.
Not used by the product path."""
import torch


def ref_spatial(x, flip, inv):
    """flip along d where flip[b], then resample: y(o) = trilinear sample at centre + inv[b] (o - centre), outside = volume minimum."""
    B, D, H, W = x.shape
    x = x.double()
    x = torch.where(flip.view(-1, 1, 1, 1).bool(), x.flip(1), x)
    fill = x.amin(dim=(1, 2, 3))
    dev = x.device
    d, h, w = torch.meshgrid(torch.arange(D, device=dev), torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    c = torch.stack([d + 0.5 - D / 2, h + 0.5 - H / 2, w + 0.5 - W / 2], 0).double().view(1, 3, -1)          # [1,3,S]
    s = inv.double() @ c + torch.tensor([D / 2 - 0.5, H / 2 - 0.5, W / 2 - 0.5], device=dev, dtype=torch.float64).view(1, 3, 1)
    f = torch.floor(s)
    t = s - f
    f = f.long()
    out = torch.zeros(B, D * H * W, dtype=torch.float64, device=dev)
    flat = x.reshape(B, -1)
    for a in (0, 1):
        for b in (0, 1):
            for cc in (0, 1):
                dd, hh, ww = f[:, 0] + a, f[:, 1] + b, f[:, 2] + cc
                inside = (dd >= 0) & (dd < D) & (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
                idx = (dd.clamp(0, D - 1) * H + hh.clamp(0, H - 1)) * W + ww.clamp(0, W - 1)
                val = torch.where(inside, flat.gather(1, idx), fill.view(-1, 1).expand_as(idx))
                wgt = (t[:, 0] if a else 1 - t[:, 0]) * (t[:, 1] if b else 1 - t[:, 1]) * (t[:, 2] if cc else 1 - t[:, 2])
                out += val * wgt
    return out.view(B, D, H, W)


def ref_blur(x, sigma, radius=8):
    """separable Gaussian, std sigma[axis][b], taps -r..r (r = min(radius, extent)), symmetric borders, renormalised taps."""
    out = x.double()
    B = x.shape[0]
    for axis in range(3):
        n = out.shape[axis + 1]
        r = min(radius, n)
        tt = torch.arange(-r, r + 1, device=x.device, dtype=torch.float64).view(1, -1)
        k = torch.exp(-0.5 * (tt / sigma[axis].double().clamp_min(1e-3).view(-1, 1)) ** 2)
        k = k / k.sum(dim=1, keepdim=True)
        pad_lo = out.narrow(axis + 1, 0, r).flip(axis + 1)
        pad_hi = out.narrow(axis + 1, n - r, r).flip(axis + 1)
        padded = torch.cat([pad_lo, out, pad_hi], dim=axis + 1)
        acc = torch.zeros_like(out)
        shape = [B, 1, 1, 1]
        for j in range(2 * r + 1):
            acc += padded.narrow(axis + 1, j, n) * k[:, j].view(shape)
        out = acc
    return out


def ref_gamma(v, gamma):
    v = v.double()
    return torch.sign(v) * torch.abs(v) ** gamma.double().view(-1, 1, 1, 1)


def ref_swap(x, origins, patch=(8, 4, 4)):
    """`origins` int [iters, B, 2, 3]; exchanges applied in order."""
    B, D, H, W = x.shape
    pd, ph, pw = patch
    out = x.clone()
    for it in range(origins.shape[0]):
        for b in range(B):
            (a0, a1, a2), (b0, b1, b2) = origins[it, b, 0].tolist(), origins[it, b, 1].tolist()
            if (a0, a1, a2) == (b0, b1, b2):
                continue
            pa = out[b, a0:a0 + pd, a1:a1 + ph, a2:a2 + pw].clone()
            pb = out[b, b0:b0 + pd, b1:b1 + ph, b2:b2 + pw].clone()
            out[b, a0:a0 + pd, a1:a1 + ph, a2:a2 + pw] = pb
            out[b, b0:b0 + pd, b1:b1 + ph, b2:b2 + pw] = pa
    return out


def ref_znorm(x):
    x = x.double()
    m = x.mean(dim=(1, 2, 3), keepdim=True)
    s = x.std(dim=(1, 2, 3), keepdim=True)            # unbiased, as torch.Tensor.std in torchio.ZNormalization
    return (x - m) / s.clamp_min(1e-12)
