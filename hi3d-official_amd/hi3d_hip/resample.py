"""Separable image resampling for the conditioner's two resizes, as banded tap tables for `hi3d_resample_axis`.

Replaces (reference call sites):
  * sgm/modules/encoders/modules.py:619-625   kornia.geometry.resize(x, (224, 224), interpolation="bicubic",
        align_corners=True, antialias=True)  -- kornia 0.6.9 (environments.yaml:107; the package is absent from this image,
        its published algorithm is restated): for a down-scale, a separable Gaussian blur (sigma = (factor - 1) / 2,
        kernel size int(max(4 sigma, 3)) made odd, 'reflect' border) followed by torch's bicubic interpolation
        (cubic convolution A = -0.75, align_corners, indices clamped to the border);
  * vtdm/encoders.py:80                        F.interpolate(y, [224, 384], mode="bilinear")[:, :, :, 80:304]

Both are linear and separable: out = R_h . X . R_w^T per channel, R = (interpolation matrix) . (blur matrix).  The rows of R
are short (<= 4 + kernel - 1 contiguous inputs, borders folded in), so each axis is one pass of a banded kernel:
out[o] = sum_t w[o][t] * in[start[o] + t].  The tables are built once per geometry on the host in float64.
"""
import math

import torch

_TABLES = {}


def _cubic_weights(t, A=-0.75):
    """torch's cubic convolution coefficients (aten UpSample.h: get_cubic_upsample_coefficients) for offsets -1, 0, 1, 2."""
    def c1(x):        # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def c2(x):        # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]


def interp_matrix(n_in, n_out, mode, align_corners):
    """[n_out, n_in] float64 matrix of F.interpolate along one axis (mode 'bilinear' / 'bicubic')."""
    M = torch.zeros((n_out, n_in), dtype=torch.float64)
    for o in range(n_out):
        if align_corners:
            src = o * (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        else:
            src = (o + 0.5) * n_in / n_out - 0.5
        if mode == "bilinear":
            if not align_corners:
                src = max(src, 0.0)                       # aten: area_pixel_compute_source_index clamps at 0
            i0 = min(int(math.floor(src)), n_in - 1)
            i1 = min(i0 + 1, n_in - 1)
            t = src - i0
            M[o, i0] += 1.0 - t
            M[o, i1] += t
        elif mode == "bicubic":
            i0 = int(math.floor(src))
            t = src - i0
            for k, wk in enumerate(_cubic_weights(t)):
                M[o, min(max(i0 - 1 + k, 0), n_in - 1)] += wk          # upsample_get_value_bounded: clamped access
        else:
            raise ValueError(mode)
    return M


def gaussian_blur_matrix(n, ksize, sigma):
    """[n, n] float64: kornia 0.6.9 gaussian_blur2d along one axis (filters/kernels.py gaussian(): x = arange(k) - k // 2,
    exp(-x^2 / (2 sigma^2)), normalised; border_type 'reflect')."""
    xs = torch.arange(ksize, dtype=torch.float64) - ksize // 2
    if ksize % 2 == 0:
        xs = xs + 0.5
    g = torch.exp(-xs * xs / (2.0 * sigma * sigma))
    g = g / g.sum()
    r = ksize // 2
    G = torch.zeros((n, n), dtype=torch.float64)
    for i in range(n):
        for k in range(ksize):
            j = i + k - r
            if j < 0:
                j = -j                                   # reflect (no edge repeat), F.pad(mode="reflect")
            if j > n - 1:
                j = 2 * (n - 1) - j
            G[i, j] += g[k]
    return G


def kornia_axis_matrices(h, w, size, interpolation="bicubic", align_corners=True, antialias=True):
    """(R_h [size0, h], R_w [size1, w]) with the blur folded in; the blur is applied on BOTH axes as soon as either
    factor exceeds 1 (kornia: `antialias and max(factors) > 1`), each with its own sigma / kernel size."""
    fh, fw = h / size[0], w / size[1]
    Rh = interp_matrix(h, size[0], interpolation, align_corners)
    Rw = interp_matrix(w, size[1], interpolation, align_corners)
    if antialias and max(fh, fw) > 1:
        sig = (max((fh - 1.0) / 2.0, 0.001), max((fw - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        Rh = Rh @ gaussian_blur_matrix(h, ks[0], sig[0])
        Rw = Rw @ gaussian_blur_matrix(w, ks[1], sig[1])
    return Rh, Rw


def band(M):
    """Dense [n_out, n_in] -> (start int32 [n_out], weights fp32 [n_out, ntap]) with every row's non-zeros inside
    [start, start + ntap)."""
    n_out, n_in = M.shape
    nz = M != 0
    first = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(n_out, dtype=torch.long))
    last = n_in - 1 - nz.flip(1).float().argmax(1)
    ntap = int((last - first).max().item()) + 1
    start = torch.minimum(first, torch.full_like(first, n_in - ntap)).clamp_(min=0)
    idx = start[:, None] + torch.arange(ntap)[None, :]
    w = torch.gather(M, 1, idx.clamp(max=n_in - 1))
    w = torch.where(idx < n_in, w, torch.zeros_like(w))
    return start.to(torch.int32).contiguous(), w.to(torch.float32).contiguous()


def tables(kind, h, w, device):
    """Cached device-side tap tables: ((start_h, w_h), (start_w, w_w), (Ho, Wo)).
    kind 'clip224'  : kornia resize to 224 x 224, bicubic + antialias, align_corners (FrozenOpenCLIPImageEmbedder.preprocess)
    kind 'clip224_noaa' : the same without the Gaussian pre-blur (antialias=False: plain bicubic taps)
    kind 'aes'      : bilinear to 224 x 384, columns 80:304 kept (AesEmbedder.forward)"""
    key = (kind, h, w, str(device))
    if key not in _TABLES:
        if kind == "clip224":
            Rh, Rw = kornia_axis_matrices(h, w, (224, 224))
        elif kind == "clip224_noaa":
            Rh, Rw = kornia_axis_matrices(h, w, (224, 224), antialias=False)
        elif kind == "aes":
            Rh = interp_matrix(h, 224, "bilinear", False)
            Rw = interp_matrix(w, 384, "bilinear", False)[80:304]
        else:
            raise ValueError(kind)
        (sh, wh), (sw, ww) = band(Rh), band(Rw)
        _TABLES[key] = ((sh.to(device), wh.to(device)), (sw.to(device), ww.to(device)), (Rh.shape[0], Rw.shape[0]))
    return _TABLES[key]
