"""Property-based GPU parity (hypothesis, derandomized): the exact paths beside stage A on random shapes and contents -
the Pillow-exact resize, the patch gather with caller-given positions, the per-clip segment means - and the optical flow
on small / odd frames (few pyramid levels, short rows) against the oracle at the tolerance of tests/test_gpu_flow.py."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st
from PIL import Image

from oracle import flow_ref, fragment_ref
from tests.gpu_common import engine

pytestmark = pytest.mark.gpu
COMMON = dict(deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@settings(max_examples=60, **COMMON)
@given(h=st.integers(224, 700), w=st.integers(224, 900), n=st.integers(1, 2), kind=st.sampled_from(["noise", "flat", "steps"]),
       seed=st.integers(0, 2 ** 31 - 1))
def test_resize_matches_pillow_on_random_sizes(h, w, n, kind, seed):
    """Every (H, W) >= 224 gives its own coefficient tables (support, 8-bit fixed-point rounding): bit-exact against Pillow."""
    g = np.random.default_rng(seed)
    if kind == "noise":
        frames = g.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    elif kind == "flat":
        frames = np.full((n, h, w, 3), int(g.integers(0, 256)), np.uint8)
    else:
        frames = (g.integers(0, 2, (n, h // 7 + 1, w // 5 + 1, 3)) * 255).astype(np.uint8).repeat(7, 1).repeat(5, 2)[:, :h, :w]
        frames = np.ascontiguousarray(frames)
    bil, lan = engine().resize_frames(torch.from_numpy(frames).cuda())
    for i in range(n):
        img = Image.fromarray(frames[i])
        assert np.array_equal(bil[i].cpu().numpy(), np.asarray(img.resize((224, 224), Image.BILINEAR))), "bilinear"
        assert np.array_equal(lan[i].cpu().numpy(), np.asarray(img.resize((224, 224), Image.LANCZOS))), "lanczos"


@settings(max_examples=80, **COMMON)
@given(h=st.integers(16, 200), w=st.integers(16, 260), t=st.integers(1, 3), seed=st.integers(0, 2 ** 31 - 1))
def test_gather_patches_with_random_positions(h, w, t, seed):
    """relax_gather_patches copies the caller's patches (any order, repeats allowed) to the tiles of the canvas; the rest stays 0."""
    g = np.random.default_rng(seed)
    imgs = g.integers(0, 256, (t, h, w, 3), dtype=np.uint8)
    ph, pw = h // 16, w // 16
    counts = g.integers(0, min(196, 3 * ph * pw) + 1, t).astype(np.int32)
    pos = np.full((t, 196, 2), -1, np.int32)
    for i in range(t):
        pos[i, :counts[i], 0] = g.integers(0, ph, counts[i])
        pos[i, :counts[i], 1] = g.integers(0, pw, counts[i])
    got = engine().gather_patches(torch.from_numpy(imgs).cuda(), torch.from_numpy(pos), torch.from_numpy(counts)).cpu().numpy()
    for i in range(t):
        assert np.array_equal(got[i], fragment_ref.gather_patches(imgs[i], pos[i, :counts[i]]))


@settings(max_examples=60, **COMMON)
@given(counts=st.lists(st.integers(1, 40), min_size=1, max_size=70), cols=st.integers(1, 300), col0=st.integers(0, 17),
       row0=st.integers(0, 5), seed=st.integers(0, 2 ** 31 - 1))
def test_segment_means_on_random_segmentations(counts, cols, col0, row0, seed):
    """relax_segment_mean: per-clip means of row blocks written into a column window of the [clips, F] matrix (more than 64 clips
    take several launches); against float64 means."""
    g = np.random.default_rng(seed)
    n = int(sum(counts))
    src = torch.from_numpy(g.standard_normal((row0 + n + 3, cols)).astype(np.float32) * 10).cuda()
    out = torch.full((len(counts), col0 + cols + 2), 7.0, device="cuda")
    engine()._segment_means(out, [(src, row0, col0)], counts)
    want = np.stack([src[row0 + a: row0 + b].double().mean(0).cpu().numpy()
                     for a, b in zip(np.cumsum([0] + counts[:-1]), np.cumsum(counts))])
    got = out.cpu().numpy()
    assert np.allclose(got[:, col0:col0 + cols], want, rtol=1e-5, atol=1e-5)
    assert (got[:, :col0] == 7.0).all() and (got[:, col0 + cols:] == 7.0).all(), "wrote outside its column window"


def _smooth_pair(h, w, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.zeros((h, w, 3), np.float32)
    for _ in range(8):
        fx, fy, ph = g.uniform(0.02, 0.2), g.uniform(0.02, 0.2), g.uniform(0, 6.28, 3)
        for c in range(3):
            base[..., c] += g.uniform(10, 30) * np.sin(xx * fx + yy * fy + ph[c])
    a = np.clip(base + 128, 0, 255).astype(np.uint8)
    b = np.roll(np.roll(a, 1, axis=1), 1, axis=0)
    return a, b


@settings(max_examples=25, **COMMON)
@given(h=st.integers(16, 140), w=st.integers(16, 180), pairs=st.integers(1, 2), seed=st.integers(0, 2 ** 31 - 1))
def test_flow_on_small_and_odd_frames(h, w, pairs, seed):
    """Frames from 16 pixels up: 1 to 3 pyramid levels instead of 4, rows shorter than a workgroup, odd sizes, blur / box windows wider
    than the image (border replication everywhere)."""
    ab = [_smooth_pair(h, w, seed + i) for i in range(pairs)]
    frames = np.stack([np.stack(p) for p in ab])
    flow, _ = engine().optical_flow(torch.from_numpy(frames).cuda(), want_flow=True, want_image=False)
    for i, (a, b) in enumerate(ab):
        want = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
        err = np.abs(flow[i].cpu().numpy() - want)
        assert err.max() < 2e-3 and err.mean() < 2e-5, (h, w, err.max(), err.mean())
