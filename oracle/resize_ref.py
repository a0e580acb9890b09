"""Oracle (TEST INFRASTRUCTURE, not product): numpy restatement of the whole-frame resize the reference applies
before the backbones.

Reference call sites (/root/reference/src):
  extractor/visualise_resnet.py:40-47   transforms.Resize((224,224)) on a PIL image = PIL BILINEAR, antialiased
  extractor/visualise_vit_layer.py:466-469  img.resize((224,224), Image.Resampling.LANCZOS)
The arithmetic lives in the third-party Pillow (pillow==10.2.0, requirements.txt:80; libImaging/Resample.c, 8 bits
per channel path): separable, horizontal pass first, 22-bit fixed-point coefficients, rounding via +2^21 then >>22,
clamped to [0,255], the horizontal result stored as uint8 before the vertical pass.  Pillow is installed in this image
(12.x, same algorithm), so this restatement is pinned directly against ``PIL.Image.resize`` in
tests/test_oracle_resize.py.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, LANCZOS = 0, 1
_SUPPORT = {BILINEAR: 1.0, LANCZOS: 3.0}


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


_FILTER = {BILINEAR: _bilinear, LANCZOS: _lanczos}


def precompute_coeffs(in_size, out_size, filt):
    """-> (bounds int32 [out,2] (xmin, count), coeffs int32 [out, ksize]) exactly as Pillow's precompute_coeffs +
    normalize_coeffs_8bpc for box = (0, in_size)."""
    f, support0 = _FILTER[filt], _SUPPORT[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        n = xmax - xmin
        k = [f((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = sum(k)           # same left-to-right double accumulation as the C loop
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            coeffs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, n)
    return bounds, coeffs


def _resample_axis0(img, out_size, filt):
    """Resample along axis 0 of a uint8 array [L, ...]."""
    bounds, coeffs = precompute_coeffs(img.shape[0], out_size, filt)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        k = coeffs[xx, :n].astype(np.int64).reshape((n,) + (1,) * (img.ndim - 1))
        acc = (1 << (PRECISION_BITS - 1)) + (src[xmin:xmin + n] * k).sum(axis=0)
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize(img_u8, out_h=224, out_w=224, filt=BILINEAR):
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]; horizontal pass first, then vertical, like ImagingResampleInner.
    A pass is skipped when that dimension already matches (Pillow copies instead)."""
    x = img_u8
    if x.shape[1] != out_w:
        x = np.ascontiguousarray(_resample_axis0(np.ascontiguousarray(x.transpose(1, 0, 2)), out_w, filt).transpose(1, 0, 2))
    if x.shape[0] != out_h:
        x = _resample_axis0(x, out_h, filt)
    return x
