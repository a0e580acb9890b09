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


# ---- contraction kernels on random geometries -----------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402

from relax_vqa_amd.engine import pack_conv_weight  # noqa: E402
from tests.gpu_common import assert_close  # noqa: E402


def _randn(g, *shape, scale=1.0):
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


@settings(max_examples=60, **COMMON)
@given(nimg=st.integers(1, 4), h=st.integers(3, 30), w=st.integers(3, 30), cin=st.sampled_from([32, 64, 96, 128, 256]),
       cout=st.sampled_from([64, 128, 192, 256, 512]), k=st.sampled_from([1, 3]), stride=st.sampled_from([1, 2]),
       act=st.sampled_from([0, 1]), with_res=st.booleans(), precision=st.sampled_from(["bf16x6", "fp32"]), seed=st.integers(0, 2 ** 31 - 1))
def test_conv2d_on_random_geometries(nimg, h, w, cin, cout, k, stride, act, with_res, precision, seed):
    """Implicit-GEMM convolution on random image sizes (rows of a tile straddle images, every tap mask pattern at the borders,
    non-square maps, row counts far from a multiple of the tile), channel counts of every tile variant, stride 1 / 2, 1x1 and 3x3,
    with and without residual: against an fp64 convolution."""
    pad = 1 if k == 3 else 0
    if precision == "fp32" and k == 3 and cin & (cin - 1):
        return      # the exact-fp32 kernel takes power-of-two channel counts for KHxKW > 1 (it says so: tests/test_gpu_errors_and_dist.py)
    g = np.random.default_rng(seed)
    x = _randn(g, nimg, cin, h, w)
    wt = _randn(g, cout, cin, k, k, scale=(cin * k * k) ** -0.5)
    b = _randn(g, cout)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad)
    res = _randn(g, *ref.shape) if with_res else None
    if with_res:
        ref = ref + res.double()
    if act == 1:
        ref = F.relu(ref)
    eng = engine()
    eng.set_precision(precision)
    got = eng.op_conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), torch.from_numpy(pack_conv_weight(wt.numpy())).cuda(), b.cuda(),
                             res.permute(0, 2, 3, 1).contiguous().cuda() if with_res else None, cout, k, k, stride, pad, act=act)
    assert_close(got.permute(0, 3, 1, 2), ref.float().numpy(), f"{precision} conv {nimg}x{h}x{w}x{cin}->{cout} k{k}s{stride} act{act} res{with_res}")


@settings(max_examples=60, **COMMON)
@given(m=st.integers(1, 1500), n=st.sampled_from([64, 128, 192, 256, 320, 768]), k=st.sampled_from([16, 32, 48, 64, 160, 768, 1024]),
       act=st.sampled_from([0, 1, 2]), with_bias=st.booleans(), with_res=st.booleans(), precision=st.sampled_from(["bf16x6", "fp32"]),
       seed=st.integers(0, 2 ** 31 - 1))
def test_gemm_on_random_shapes(m, n, k, act, with_bias, with_res, precision, seed):
    """out = act(A W^T + bias + residual) on random row counts (partial tiles, fewer rows than a tile, tail split-K) and every
    tile variant, against an fp64 product."""
    if precision == "fp32" and k % 32:
        return
    g = np.random.default_rng(seed)
    A, W = _randn(g, m, k), _randn(g, n, k, scale=k ** -0.5)
    b = _randn(g, n) if with_bias else None
    r = _randn(g, m, n) if with_res else None
    y = A.double() @ W.double().T
    if with_bias:
        y = y + b.double()
    if with_res:
        y = y + r.double()
    want = [y, F.relu(y), F.gelu(y)][act].float().numpy()
    eng = engine()
    eng.set_precision(precision)
    got = eng.op_gemm(A.cuda(), W.cuda(), b.cuda() if with_bias else None, r.cuda() if with_res else None, act=act)
    assert_close(got, want, f"{precision} gemm {m}x{n}x{k} act{act} bias{with_bias} res{with_res}")


@settings(max_examples=30, **COMMON)
@given(n_img=st.integers(1, 5), heads=st.sampled_from([1, 3, 6, 12]), scale=st.floats(0.05, 5.0), outlier=st.booleans(),
       precision=st.sampled_from(["bf16x6", "fp32"]), seed=st.integers(0, 2 ** 31 - 1))
def test_attention_on_random_inputs(n_img, heads, scale, outlier, precision, seed):
    """softmax(q k^T / 8) v for random image / head counts and logit magnitudes from near-uniform to near one-hot rows; with
    `outlier` one query and one key carry 6x larger entries (dominant logits of a few hundred: the max-subtraction path; much larger
    ones lose digits in ANY fp32 evaluation of the logit itself - 30x was tried: 1.04x the elementwise bar on the exact-fp32 path)."""
    g = np.random.default_rng(seed)
    dim = heads * 64
    qkv = _randn(g, n_img * 197, 3 * dim, scale=scale)
    if outlier:
        qkv[int(g.integers(0, n_img * 197)), :dim] *= 6.0
        qkv[int(g.integers(0, n_img * 197)), dim:2 * dim] *= 6.0
    t = qkv.double().reshape(n_img, 197, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (((t[0] @ t[1].transpose(-2, -1)) * 64 ** -0.5).softmax(dim=-1) @ t[2]).transpose(1, 2).reshape(n_img * 197, dim)
    eng = engine()
    eng.set_precision(precision)
    got = eng.op_attention(qkv.cuda(), n_img, heads)
    assert_close(got, ref.float().numpy(), f"{precision} attention n={n_img} heads={heads} scale={scale:.2f} outlier={outlier}")


@settings(max_examples=40, **COMMON)
@given(rows=st.integers(1, 700), dim=st.sampled_from([64, 192, 384, 768]), spread=st.floats(1e-3, 1e3), offset=st.floats(-100, 100),
       seed=st.integers(0, 2 ** 31 - 1))
def test_layernorm_on_random_rows(rows, dim, spread, offset, seed):
    """LayerNorm (eps 1e-6) on rows of any scale and offset, gammas of mixed sign, against fp64."""
    g = np.random.default_rng(seed)
    x = _randn(g, rows, dim, scale=spread) + np.float32(offset)
    gamma, beta = _randn(g, dim), _randn(g, dim)
    ref = F.layer_norm(x.double(), (dim,), gamma.double(), beta.double(), 1e-6)
    got = engine().op_layernorm(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6)
    # a large offset against a small spread loses digits in ANY fp32 evaluation of (x - mean): compare at that resolution
    tol = max(1e-4, 4e-7 * (abs(offset) + spread) / spread)
    assert np.allclose(got.cpu().numpy(), ref.float().numpy(), rtol=tol, atol=tol * 3), (rows, dim, spread, offset)


# ---- whole clips: ragged batches ----------------------------------------------------------------------------------------
from tests.gpu_common import rn50_weights, synth, vit_weights  # noqa: E402


@settings(max_examples=8, **COMMON)
@given(shapes=st.lists(st.tuples(st.integers(1, 5), st.integers(16, 300), st.integers(16, 400)), min_size=1, max_size=4),
       seed=st.integers(0, 10 ** 6))
def test_clip_vectors_on_ragged_batches(shapes, seed):
    """Clips with different numbers of pairs and different frame sizes (down to a single patch) in ONE batched pass: with the tail
    split-K off every clip's 19779-d vector is bit-identical to running the clip alone; the first clip is also checked against
    the oracle pipeline (fragments -> ResNet-50 / ViT features -> per-clip mean)."""
    from oracle import pipeline_ref
    rn_sd = rn50_weights()
    vit_sd = vit_weights("vit_base")
    eng = engine()
    clips_np = [synth.synthetic_clip(t, h, w, clip_id=seed + i) for i, (t, h, w) in enumerate(shapes)]
    clips = [torch.from_numpy(c).cuda() for c in clips_np]
    eng.set_option("gemm_split_k", 0)
    try:
        both = eng.clip_vectors(clips)
        for i, c in enumerate(clips):
            assert torch.equal(eng.clip_vectors([c])[0], both[i]), f"clip {i} of {shapes} depends on its batch"
    finally:
        eng.set_option("gemm_split_k", 1)
    assert both.shape == (len(clips), 19779) and bool(torch.isfinite(both).all())
    if shapes[0][0] <= 2:       # the oracle runs both backbones on the CPU: only short first clips
        feats = pipeline_ref.clip_features(clips_np[0], rn_sd, vit_sd, schedule="dedup")
        want = np.concatenate([feats["resnet"], feats["vit"]], axis=1).mean(axis=0)
        got = both[0].cpu().numpy()
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-4


# ---- quality head: missing values -----------------------------------------------------------------------------------------
@settings(max_examples=25, **COMMON)
@given(n=st.integers(1, 70), feats=st.sampled_from([256, 4608, 19779]), hidden=st.sampled_from([128, 256, 512]), bad=st.floats(0.0, 0.3),
       seed=st.integers(0, 2 ** 31 - 1))
def test_head_with_random_nan_patterns(n, feats, hidden, bad, seed):
    """NaN -> imputer mean -> MinMaxScaler -> MLP (src/demo_test.py:177-208) for random batch sizes, input widths and fractions of missing
    entries (up to whole rows), against the oracle restatement of the reference's head.  (+-inf is outside the reference's domain:
    its SimpleImputer raises on it.)"""
    from oracle import mlp_ref
    g = np.random.default_rng(seed)
    sd = synth.mlp_head_state_dict(feats, hidden, seed=seed % 1000)
    scale = g.uniform(0.01, 2.0, feats)
    mn = g.uniform(-1, 1, feats)
    imput = g.standard_normal(feats)
    x = (g.standard_normal((n, feats)) * 3).astype(np.float32)
    mask = g.random((n, feats)) < bad
    x[mask] = np.nan
    if n > 1 and bad > 0.15:
        x[0] = np.nan
    eng = engine()
    eng.load_mlp_head(sd, scale, mn, imput)
    got = eng.mlp_head(torch.from_numpy(x).cuda()).cpu().numpy()
    want = mlp_ref.predict(sd, x, imput, scale, mn)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-3 * (1 + np.abs(want).max()))


# ---- full ReLaX vectors (flow + whole frames) -----------------------------------------------------------------------------
@settings(max_examples=6, **COMMON)
@given(shapes=st.lists(st.tuples(st.integers(1, 4), st.integers(16, 260), st.integers(16, 330)), min_size=1, max_size=3),
       max_pairs=st.sampled_from([0, 1, 2, 3]), seed=st.integers(0, 10 ** 6))
def test_full_clip_vectors_on_ragged_batches_and_flow_chunking(shapes, max_pairs, seed):
    """The 35203-d vectors (whole-frame resize, Farneback flow fragments, both backbones) of a ragged batch: bit-identical to clip by
    clip with the tail split-K off, and independent of how many pairs the optical flow takes per launch (`flow_max_pairs`)."""
    rn50_weights()
    vit_weights("vit_base")
    eng = engine()
    clips = [torch.from_numpy(synth.synthetic_clip(t, h, w, clip_id=seed + i)).cuda() for i, (t, h, w) in enumerate(shapes)]
    eng.set_option("gemm_split_k", 0)
    try:
        both = eng.full_clip_vectors(clips, flow=True)
        eng.set_option("flow_max_pairs", max_pairs)
        for i, c in enumerate(clips):
            assert torch.equal(eng.full_clip_vectors([c], flow=True)[0], both[i]), f"clip {i} of {shapes} depends on its batch / flow chunking"
    finally:
        eng.set_option("gemm_split_k", 1)
        eng.set_option("flow_max_pairs", 0)
    assert both.shape == (len(clips), 35203) and bool(torch.isfinite(both).all())


def test_results_do_not_depend_on_what_the_workspaces_held():
    """`debug_poison` fills every workspace with 0xFF bytes (NaN / -1) whenever an entry point requests it.  A kernel that read
    workspace it had not written in the same call would turn that into NaNs or different bits; the vectors must not move."""
    rn50_weights()
    vit_weights("vit_base")
    eng = engine()
    clips = [torch.from_numpy(synth.synthetic_clip(t, h, w, clip_id=300 + i)).cuda() for i, (t, h, w) in enumerate([(3, 272, 400), (2, 96, 130), (1, 16, 16)])]
    a = eng.clip_vectors(clips)
    fa = eng.full_clip_vectors(clips, flow=True)
    ga = eng.op_gemm(torch.ones(700, 768, device="cuda"), torch.ones(768, 768, device="cuda"))
    eng.set_option("debug_poison", 1)
    try:
        assert eng.get_option("debug_poison") == 1
        b = eng.clip_vectors(clips)
        fb = eng.full_clip_vectors(clips, flow=True)
        gb = eng.op_gemm(torch.ones(700, 768, device="cuda"), torch.ones(768, 768, device="cuda"))
    finally:
        eng.set_option("debug_poison", 0)
    assert torch.equal(a, b) and torch.equal(fa, fb) and torch.equal(ga, gb)
    assert bool(torch.isfinite(b).all()) and bool(torch.isfinite(fb).all())
