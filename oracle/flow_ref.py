"""Oracle (TEST INFRASTRUCTURE, not product): Farneback dense optical flow + the flow visualisation of ReLaX-VQA
(SURVEY §8(a) A7-A8, §8(f) f2).

Reference call sites (/root/reference/src):
  main_fragment_layerstack.py:313-315  cv2.calcOpticalFlowFarneback(gray(orig), gray(next), None, 0.5, 3, 15, 3, 5, 1.2, 0)
  main_fragment_layerstack.py:162-175  flow_to_rgb: cartToPolar, NORM_MINMAX to [0,255], hue = ang*180/pi/2, HSV->BGR
The arithmetic lives in the third-party opencv-python==4.9.0.80 (requirements.txt:73), which is absent from the
reference tree and from this image.  This file restates OpenCV's published algorithm (modules/video/src/optflowgf.cpp:
Gaussian pyramid by blur+linear resize, polynomial expansion, matrix update, box-blurred flow update; core/imgproc:
BGR2GRAY fixed point, fastAtan2 polynomial, 8-bit HSV->BGR) in float32 numpy.  PARITY: the only pins are the reference's
example `*_residual_of.png` images (4 real pairs); float rounding differs from OpenCV's SIMD kernels, so this is a
TOLERANCE check (tests/test_oracle_flow.py), not a bit-exact one.
"""
import numpy as np

F32 = np.float32


# ---- colour / helpers ------------------------------------------------------------------------------------------
def bgr2gray(img):
    """cv2.COLOR_BGR2GRAY on uint8: (B*1868 + G*9617 + R*4899 + 2^13) >> 14."""
    b, g, r = (img[..., i].astype(np.int32) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def cv_round(x):
    return int(np.rint(x))   # round half to even, like cvRound (lrint)


def gaussian_kernel(ksize, sigma):
    """cv::getGaussianKernel(ksize, sigma, CV_32F)."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize <= 7 and ksize % 2 == 1 and sigma <= 0:
        return np.array(small[ksize], dtype=F32)
    s = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-0.5 / (s * s) * x * x).astype(F32)
    return (k * F32(1.0 / float(k.astype(np.float64).sum()))).astype(F32)


def _reflect101(idx, n):
    idx = np.abs(idx)
    return np.where(idx >= n, 2 * (n - 1) - idx, idx)


def gaussian_blur(img, ksize, sigma):
    """Separable float32 Gaussian, BORDER_REFLECT_101 (cv::GaussianBlur default)."""
    k = gaussian_kernel(ksize, sigma)
    r = ksize // 2
    h, w = img.shape
    cols = _reflect101(np.arange(-r, w + r), w)
    padded = img[:, cols]
    tmp = np.zeros_like(img)
    for i in range(ksize):
        tmp += k[i] * padded[:, i:i + w]
    rows = _reflect101(np.arange(-r, h + r), h)
    padded = tmp[rows]
    out = np.zeros_like(img)
    for i in range(ksize):
        out += k[i] * padded[i:i + h]
    return out


def resize_linear(img, out_h, out_w):
    """cv::resize(..., INTER_LINEAR) on float32 [H,W] or [H,W,C] (no antialiasing)."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()

    def taps(n_in, n_out):
        scale = n_in / n_out
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(F32)   # OpenCV: (float)((dx+0.5)*scale-0.5)
        s = np.floor(f).astype(np.int64)
        a = (f - s.astype(F32)).astype(F32)
        lo = s < 0
        s[lo], a[lo] = 0, 0
        hi = s >= n_in - 1
        s[hi], a[hi] = n_in - 1, 0
        return s, np.minimum(s + 1, n_in - 1), a

    sx0, sx1, ax = taps(w, out_w)
    sy0, sy1, ay = taps(h, out_h)
    if img.ndim == 3:
        ax_, ay_ = ax[None, :, None], ay[:, None, None]
    else:
        ax_, ay_ = ax[None, :], ay[:, None]
    hor = img[:, sx0] * (F32(1) - ax_) + img[:, sx1] * ax_
    return (hor[sy0] * (F32(1) - ay_) + hor[sy1] * ay_).astype(F32)


# ---- Farneback ---------------------------------------------------------------------------------------------------
def _prepare_gaussian(n, sigma):
    if sigma < np.finfo(np.float32).eps:
        sigma = n * 0.3
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sigma * sigma)).astype(F32)
    s = 1.0 / float(g.astype(np.float64).sum())
    g = (g.astype(np.float64) * s).astype(F32)
    xg = (x * g.astype(np.float64)).astype(F32)
    xxg = (x * x * g.astype(np.float64)).astype(F32)
    G = np.zeros((6, 6))
    gd = g.astype(np.float64)
    for yy in range(-n, n + 1):
        for xx in range(-n, n + 1):
            w = gd[yy + n] * gd[xx + n]
            G[0, 0] += w
            G[1, 1] += w * xx * xx
            G[3, 3] += w * xx * xx * xx * xx
            G[5, 5] += w * xx * xx * yy * yy
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    return g[n:], xg[n:], xxg[n:], inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]


def poly_exp(img, n=5, sigma=1.2):
    """FarnebackPolyExp: float32 [H,W] -> float32 [H,W,5] (r2..r6 in OpenCV's storage order)."""
    g, xg, xxg, ig11, ig03, ig33, ig55 = _prepare_gaussian(n, sigma)
    h, w = img.shape
    ys = np.arange(h)
    t0 = img * g[0]
    t1 = np.zeros_like(img)
    t2 = np.zeros_like(img)
    for k in range(1, n + 1):
        up = img[np.maximum(ys - k, 0)]
        dn = img[np.minimum(ys + k, h - 1)]
        p = up + dn
        t0 = t0 + g[k] * p
        t1 = t1 + xg[k] * (dn - up)
        t2 = t2 + xxg[k] * p
    cols = np.clip(np.arange(-n, w + n), 0, w - 1)      # replicated border
    r0, r1, r2 = t0[:, cols], t1[:, cols], t2[:, cols]
    c = slice(n, n + w)
    b1 = (r0[:, c] * g[0]).astype(np.float64)
    b3 = (r1[:, c] * g[0]).astype(np.float64)
    b5 = (r2[:, c] * g[0]).astype(np.float64)
    b2 = np.zeros((h, w))
    b4 = np.zeros((h, w))
    b6 = np.zeros((h, w))
    for k in range(1, n + 1):
        p, m = slice(n + k, n + k + w), slice(n - k, n - k + w)
        tg = (r0[:, p] + r0[:, m]).astype(np.float64)
        b1 += tg * float(g[k])
        b4 += tg * float(xxg[k])
        b2 += ((r0[:, p] - r0[:, m]) * xg[k]).astype(np.float64)
        b3 += ((r1[:, p] + r1[:, m]) * g[k]).astype(np.float64)
        b6 += ((r1[:, p] - r1[:, m]) * xg[k]).astype(np.float64)
        b5 += ((r2[:, p] + r2[:, m]) * g[k]).astype(np.float64)
    out = np.empty((h, w, 5), dtype=F32)
    out[..., 1] = b2 * ig11
    out[..., 0] = b3 * ig11
    out[..., 3] = b1 * ig03 + b4 * ig33
    out[..., 2] = b1 * ig03 + b5 * ig33
    out[..., 4] = b6 * ig55
    return out


_BORDER = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472], dtype=F32)


def update_matrices(R0, R1, flow):
    """FarnebackUpdateMatrices over the whole image -> float32 [H,W,5]."""
    h, w = flow.shape[:2]
    xs = np.arange(w, dtype=F32)[None, :]
    ys = np.arange(h, dtype=F32)[:, None]
    dx, dy = flow[..., 0], flow[..., 1]
    fx, fy = xs + dx, ys + dy
    x1, y1 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    fx, fy = (fx - x1).astype(F32), (fy - y1).astype(F32)
    ok = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc, yc = np.clip(x1, 0, w - 2), np.clip(y1, 0, h - 2)
    a00, a01, a10, a11 = (1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy
    samp = (a00[..., None] * R1[yc, xc] + a01[..., None] * R1[yc, xc + 1] +
            a10[..., None] * R1[yc + 1, xc] + a11[..., None] * R1[yc + 1, xc + 1]).astype(F32)
    r2 = np.where(ok, samp[..., 0], F32(0))
    r3 = np.where(ok, samp[..., 1], F32(0))
    r4 = np.where(ok, (R0[..., 2] + samp[..., 2]) * F32(0.5), R0[..., 2])
    r5 = np.where(ok, (R0[..., 3] + samp[..., 3]) * F32(0.5), R0[..., 3])
    r6 = np.where(ok, (R0[..., 4] + samp[..., 4]) * F32(0.25), R0[..., 4] * F32(0.5))
    r2 = (R0[..., 0] - r2) * F32(0.5)
    r3 = (R0[..., 1] - r3) * F32(0.5)
    r2 = r2 + r4 * dy + r6 * dx
    r3 = r3 + r6 * dy + r5 * dx
    sx = np.ones(w, dtype=F32)
    sy = np.ones(h, dtype=F32)
    for i in range(min(5, w)):
        sx[i] *= _BORDER[i]
        sx[w - 1 - i] *= _BORDER[i]
    for i in range(min(5, h)):
        sy[i] *= _BORDER[i]
        sy[h - 1 - i] *= _BORDER[i]
    scale = (sy[:, None] * sx[None, :]).astype(F32)
    r2, r3, r4, r5, r6 = (v * scale for v in (r2, r3, r4, r5, r6))
    M = np.empty((h, w, 5), dtype=F32)
    M[..., 0] = r4 * r4 + r6 * r6
    M[..., 1] = (r4 + r5) * r6
    M[..., 2] = r5 * r5 + r6 * r6
    M[..., 3] = r4 * r2 + r6 * r3
    M[..., 4] = r6 * r2 + r5 * r3
    return M


def update_flow_blur(M, block_size=15):
    """FarnebackUpdateFlow_Blur: box filter (replicated border) of M in double, then the 2x2 solve -> flow [H,W,2]."""
    m = block_size // 2
    h, w = M.shape[:2]
    rows = np.clip(np.arange(-m, h + m), 0, h - 1)
    cs = np.concatenate([np.zeros((1, w, 5)), np.cumsum(M[rows].astype(np.float64), axis=0)], axis=0)
    v = cs[block_size:] - cs[:-block_size]
    cols = np.clip(np.arange(-m, w + m), 0, w - 1)
    cs = np.concatenate([np.zeros((h, 1, 5)), np.cumsum(v[:, cols], axis=1)], axis=1)
    b = (cs[:, block_size:] - cs[:, :-block_size]) * (1.0 / (block_size * block_size))
    g11, g12, g22, h1, h2 = (b[..., i] for i in range(5))
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    flow = np.empty((h, w, 2), dtype=F32)
    flow[..., 0] = (g11 * h2 - g12 * h1) * idet
    flow[..., 1] = (g22 * h1 - g12 * h2) * idet
    return flow


def farneback(prev_gray, next_gray, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2):
    """cv2.calcOpticalFlowFarneback(prev, next, None, .5, 3, 15, 3, 5, 1.2, 0) -> float32 [H,W,2]."""
    h0, w0 = prev_gray.shape
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if w0 * scale < 32 or h0 * scale < 32:
            break
        k += 1
    levels = k
    imgs = [prev_gray.astype(F32), next_gray.astype(F32)]
    prev_flow = None
    for k in range(levels, -1, -1):
        scale = pyr_scale ** k
        sigma = (1.0 / scale - 1) * 0.5
        smooth = max(cv_round(sigma * 5) | 1, 3)
        w, h = cv_round(w0 * scale), cv_round(h0 * scale)
        if prev_flow is None:
            flow = np.zeros((h, w, 2), dtype=F32)
        else:
            flow = resize_linear(prev_flow, h, w) * F32(1.0 / pyr_scale)
        R = [poly_exp(resize_linear(gaussian_blur(im, smooth, sigma), h, w), poly_n, poly_sigma) for im in imgs]
        M = update_matrices(R[0], R[1], flow)
        for i in range(iterations):
            flow = update_flow_blur(M, winsize)
            if i < iterations - 1:
                M = update_matrices(R[0], R[1], flow)
        prev_flow = flow
    return prev_flow


# ---- visualisation (flow_to_rgb) ---------------------------------------------------------------------------------------
def fast_atan2_deg(y, x):
    """cv::fastAtan2 (degrees), the polynomial cartToPolar uses."""
    p1 = F32(0.9997878412794807 * (180 / np.pi))
    p3 = F32(-0.3258083974640975 * (180 / np.pi))
    p5 = F32(0.1555786518463281 * (180 / np.pi))
    p7 = F32(-0.04432655554792128 * (180 / np.pi))
    ax, ay = np.abs(x), np.abs(y)
    eps = F32(2.220446049250313e-16)
    big = ax >= ay
    c = np.where(big, ay / (ax + eps), ax / (ay + eps)).astype(F32)
    c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(big, a, F32(90) - a)
    a = np.where(x < 0, F32(180) - a, a)
    a = np.where(y < 0, F32(360) - a, a)
    return a.astype(F32)


def _normalize_minmax(a, lo=0.0, hi=255.0):
    """cv2.normalize(a, None, lo, hi, NORM_MINMAX) on float32."""
    smin, smax = float(a.min()), float(a.max())
    scale = (hi - lo) * (1.0 / (smax - smin) if smax - smin > np.finfo(np.float64).eps else 0.0)
    shift = lo - smin * scale
    return (a * F32(scale) + F32(shift)).astype(F32)


def hsv_to_bgr_u8(hsv):
    """cv2.cvtColor(hsv_u8, COLOR_HSV2BGR), H in [0,180): OpenCV's float formula (h*6/180, sector table,
    v*(1-s*f)) scaled by 255 and converted by TRUNCATION - established empirically against the reference's
    `*_residual_of.png` images (round-to-nearest reproduces 96.3 % of their bytes, truncation 99.98 %)."""
    hch = hsv[..., 0].astype(F32) * F32(6.0 / 180.0)
    s = hsv[..., 1].astype(F32) * F32(1.0 / 255.0)
    v = hsv[..., 2].astype(F32) * F32(1.0 / 255.0)
    sector = np.floor(hch).astype(np.int64)
    f = (hch - sector.astype(F32)).astype(F32)
    sector = np.mod(sector, 6)
    t1 = v * (F32(1) - s)
    t2 = v * (F32(1) - s * f)
    t3 = v * (F32(1) - s * (F32(1) - f))
    tab = np.stack([v, t1, t2, t3], axis=-1).astype(F32)
    # sector -> (b, g, r) indices into tab: OpenCV's sector_data
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    bgr = np.take_along_axis(tab, sd[sector], axis=-1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr)
    return np.clip(np.floor(bgr * F32(255)), 0, 255).astype(np.uint8)


def flow_to_rgb(flow):
    """src/main_fragment_layerstack.py:162-175 (returns BGR despite the name)."""
    x, y = flow[..., 0], flow[..., 1]
    mag = np.sqrt(x * x + y * y).astype(F32)
    ang = (fast_atan2_deg(y, x) * F32(np.pi / 180)).astype(F32)
    mag = _normalize_minmax(mag)
    hue = ang * F32(180) / F32(np.pi) / F32(2)
    hsv = np.zeros(flow.shape[:2] + (3,), dtype=np.uint8)
    hsv[..., 0] = hue.astype(np.uint8)          # numpy float->uint8 assignment truncates
    hsv[..., 1] = 255
    hsv[..., 2] = _normalize_minmax(mag).astype(np.uint8)
    return hsv_to_bgr_u8(hsv)
