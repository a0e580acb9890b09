"""Optical flow (Farneback + flow_to_rgb) on the GPU: against the oracle (tight float tolerance) and against the
reference's own OpenCV output on the shipped 960x540 pair (tolerance pin)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import flow_ref, fragment_ref
from tests.gpu_common import engine

pytestmark = pytest.mark.gpu


def _load(golden_dir, suffix):
    p = os.path.join(golden_dir, "png_5636101558_3", f"5636101558_3{suffix}.png")
    return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1])


def _smooth_pair(h, w, seed):
    """A textured frame and a slightly warped copy (random noise has no meaningful flow)."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.zeros((h, w, 3), np.float32)
    for _ in range(12):
        fx, fy, ph = g.uniform(0.01, 0.12), g.uniform(0.01, 0.12), g.uniform(0, 6.28, 3)
        for c in range(3):
            base[..., c] += g.uniform(10, 30) * np.sin(xx * fx + yy * fy + ph[c])
    a = np.clip(base + 128, 0, 255).astype(np.uint8)
    b = np.roll(np.roll(a, 2, axis=1), 1, axis=0)
    b[h // 3: 2 * h // 3] = np.roll(a, -3, axis=1)[h // 3: 2 * h // 3]
    return a, b


def test_flow_matches_oracle():
    a, b = _smooth_pair(200, 264, 1)
    frames = np.stack([a, b])[None]
    flow, img = engine().optical_flow(torch.from_numpy(frames).cuda(), want_flow=True, want_image=True)
    flow, img = flow[0].cpu().numpy(), img[0].cpu().numpy()
    want = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
    err = np.abs(flow - want)
    assert err.max() < 1e-3 and err.mean() < 1e-5, (err.max(), err.mean())     # float reassociation only (measured 3e-5 / 6e-7)
    want_img = flow_ref.flow_to_rgb(want)
    d = np.abs(img.astype(np.int32) - want_img.astype(np.int32))
    assert (d == 0).mean() > 0.995 and (d <= 1).mean() > 0.9995, ((d == 0).mean(), (d <= 1).mean())
    # visualisation alone on the oracle's flow: exact input, so (near) exact output
    img2 = engine().flow_to_rgb(torch.from_numpy(want[None]).cuda())[0].cpu().numpy()
    assert (img2 == want_img).mean() > 0.999      # hue / value truncation boundaries under FMA contraction


@pytest.mark.parametrize("h,w", [(540, 960), (1080, 1920)])
def test_flow_matches_oracle_at_video_sizes(h, w):
    """Many column bands and row segments of the fused kernels, 16-byte paths, all four pyramid levels: the flow itself
    (not only its visualisation) against the oracle at the sizes of BASELINE configs 3 and 4."""
    a, b = _smooth_pair(h, w, 3)
    flow, _ = engine().optical_flow(torch.from_numpy(np.stack([a, b])[None]).cuda(), want_flow=True, want_image=False)
    want = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
    err = np.abs(flow[0].cpu().numpy() - want)
    assert err.max() < 2e-3 and err.mean() < 1e-5, (err.max(), err.mean())


@pytest.mark.parametrize("h,w", [(97, 131), (150, 203), (136, 240)])
def test_flow_odd_sizes_take_the_general_kernels(h, w):
    """Row lengths that are not a multiple of 4 (and odd pixel counts) cannot use the 16-byte-load kernels: same tolerance."""
    a, b = _smooth_pair(h, w, 5)
    flow, _ = engine().optical_flow(torch.from_numpy(np.stack([a, b])[None]).cuda(), want_flow=True, want_image=False)
    want = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
    err = np.abs(flow[0].cpu().numpy() - want)
    assert err.max() < 1e-3 and err.mean() < 1e-5, (err.max(), err.mean())     # measured <= 2e-5 / 4e-7


def test_reference_png_pair(golden_dir):
    orig, nxt, want = _load(golden_dir, ""), _load(golden_dir, "_next"), _load(golden_dir, "_residual_of")
    _, img = engine().optical_flow(torch.from_numpy(np.stack([orig, nxt])[None]).cuda())
    img = img[0].cpu().numpy()
    d = np.abs(img.astype(np.int32) - want.astype(np.int32))
    # measured 0.99903 / 0.999992 (the oracle itself reproduces 99.93-99.98 % of the reference's bytes, tests/golden/pin_report.json)
    assert (d == 0).mean() > 0.998 and (d <= 1).mean() > 0.9999, ((d == 0).mean(), (d <= 1).mean())
    fo = engine().fragment_image(torch.from_numpy(img[None]).cuda())
    n = int(fo["counts"][0])
    got_pos = set(map(tuple, fo["positions"][0, :n].cpu().numpy().tolist()))
    _, wp = fragment_ref.extract_important_patches(want, fragment_ref.get_patch_diff(want))
    assert len(got_pos & set(map(tuple, wp.tolist()))) >= 195
    # and the whole merged-fragment step of the reference (src/main_fragment_layerstack.py:319-325)
    fr = engine().fragment_pairs(torch.from_numpy(np.stack([orig, nxt])[None]).cuda())
    merged = engine().merge_fragments(fr["diff_frag"], fo["frag"])[0].cpu().numpy()
    ref_merged = _load(golden_dir, "_residual_merged_frag")
    assert (merged == ref_merged).mean() > 0.995


@pytest.mark.parametrize("h,w,t", [(200, 264, 3), (97, 131, 2), (540, 960, 2)])
def test_flow_image_of_the_fused_range_equals_flow_to_rgb_of_the_flow(h, w, t):
    """relax_optical_flow takes the per-pair magnitude range for the visualisation while its last solve writes the flow
    (box_solve_fused<true>); relax_flow_to_rgb on the returned flow takes it with a pass of its own (mag_minmax).  Same
    expression on the same values: the two images are equal byte for byte, for every pair of a batch."""
    frames = np.stack([np.stack(_smooth_pair(h, w, 20 + i)) for i in range(t)])
    eng = engine()
    flow, img = eng.optical_flow(torch.from_numpy(frames).cuda(), want_flow=True, want_image=True)
    again = eng.flow_to_rgb(flow)
    assert torch.equal(img, again)
    only = eng.optical_flow(torch.from_numpy(frames).cuda(), want_flow=False, want_image=True)[1]
    assert torch.equal(img, only)


def test_batch_of_pairs_and_chunking():
    pairs = [_smooth_pair(120, 168, s) for s in range(3)]
    frames = np.stack([np.stack(p) for p in pairs])
    flow, _ = engine().optical_flow(torch.from_numpy(frames).cuda(), want_flow=True, want_image=False)
    one, _ = engine().optical_flow(torch.from_numpy(frames[1:2]).cuda(), want_flow=True, want_image=False)
    assert torch.equal(flow[1], one[0])          # a pair's flow does not depend on its batch


@pytest.mark.parametrize("h,w", [(97, 131), (35, 49)])
def test_odd_pixel_counts_with_several_pairs(h, w):
    """h * w odd: the coefficient block of every second pair ([2][h][w][4] + [2][h][w] floats, 40 bytes per pixel) starts 8 bytes off a
    16-byte line, so its 16-byte stores (poly_expansion) and loads (the iteration kernels) are misaligned.  Every pair of a batch must
    still get the flow it gets alone, on both iteration paths."""
    frames = torch.from_numpy(np.stack([np.stack(_smooth_pair(h, w, 70 + i)) for i in range(3)])).cuda()
    eng = engine()
    for fused in (1, 0):
        eng.set_option("flow_fused", fused)
        try:
            flow, img = eng.optical_flow(frames, want_flow=True, want_image=True)
            for i in range(3):
                one, one_img = eng.optical_flow(frames[i:i + 1], want_flow=True, want_image=True)
                assert torch.equal(flow[i], one[0]) and torch.equal(img[i], one_img[0]), (fused, i)
        finally:
            eng.set_option("flow_fused", 1)
    want = flow_ref.farneback(flow_ref.bgr2gray(frames[1, 0].cpu().numpy()), flow_ref.bgr2gray(frames[1, 1].cpu().numpy()))
    err = np.abs(flow[1].cpu().numpy() - want)
    assert err.max() < 1e-3 and err.mean() < 1e-5, (err.max(), err.mean())


def test_degenerate_pairs_identical_constant_and_black_frames():
    """A frozen frame (identical pair: OpenCV's border rule still yields a small flow near the right / bottom edge), two flat frames of
    different brightness (no gradients: the flow is exactly zero, the magnitude range is empty - cv2.normalize's scale falls back to 0 -
    and the image is black) and a frame against black: against the oracle as every other pair, nothing may turn into NaN, and the
    fragment stage must order an image whose patch scores are all equal by the reference's tie rule."""
    h, w = 200, 264
    a, _ = _smooth_pair(h, w, 11)
    flat_lo = np.full((h, w, 3), 50, np.uint8)
    flat_hi = np.full((h, w, 3), 200, np.uint8)
    black = np.zeros((h, w, 3), np.uint8)
    pairs = [np.stack([a, a]), np.stack([flat_lo, flat_hi]), np.stack([a, black])]
    frames = torch.from_numpy(np.stack(pairs)).cuda()
    eng = engine()
    flow, img = eng.optical_flow(frames, want_flow=True, want_image=True)
    assert bool(torch.isfinite(flow).all())
    for i, (p0, p1) in enumerate(pairs):
        want = flow_ref.farneback(flow_ref.bgr2gray(p0), flow_ref.bgr2gray(p1))
        want_img = flow_ref.flow_to_rgb(want)
        err = np.abs(flow[i].cpu().numpy() - want)
        assert err.max() < 1e-3, (i, err.max())
        d = np.abs(img[i].cpu().numpy().astype(np.int32) - want_img.astype(np.int32))
        assert (d == 0).mean() > 0.99, (i, (d == 0).mean(), d.max())
    assert float(flow[1].abs().max()) == 0.0 and int(img[1].max()) == 0          # flat frames: exactly zero flow, black image
    flat_img = img[1].cpu().numpy()
    fo = eng.fragment_image(img[1:2])
    _, wp = fragment_ref.extract_important_patches(flat_img, fragment_ref.get_patch_diff(flat_img))
    n = int(fo["counts"][0])
    assert n == len(wp) and fo["positions"][0, :n].cpu().numpy().tolist() == wp.tolist()     # all scores equal: the reference's tie order


def test_full_relax_clip_with_flow():
    from tests.gpu_common import rn50_weights, vit_weights
    rn50_weights(), vit_weights("vit_base")
    a, b = _smooth_pair(272, 400, 9)
    frames = torch.from_numpy(np.stack([a, b])[None]).cuda()
    out = engine().extract_clip(frames, flow=True)
    assert out["resnet"].shape == (1, 15171) and out["vit"].shape == (1, 4608)
    assert bool(torch.isfinite(out["resnet"]).all()) and bool(torch.isfinite(out["vit"]).all())
    vec = engine().full_clip_vector(frames, flow=True)
    assert vec.shape == (35203,)


@pytest.mark.parametrize("h,w,t", [(200, 264, 2), (97, 131, 1), (150, 203, 2), (32, 48, 1), (16, 16, 1), (540, 960, 2), (283, 1000, 1), (1080, 1920, 1)])
def test_fused_iteration_kernel_equals_the_two_kernel_path_bit_for_bit(h, w, t):
    """flow_iteration (matrix entries + box blur + solve in one kernel, M never in HBM; the default) against update_matrices_k +
    box_solve_fused (option flow_fused = 0): the same matrix entries (shared source), the same additions in the same order - flow
    and flow image identical bit for bit, for every pair of a batch, on sizes with one to four pyramid levels, partial bands,
    short segments and rows that are not a multiple of 3 or 4."""
    frames = torch.from_numpy(np.stack([np.stack(_smooth_pair(h, w, 40 + i)) for i in range(t)])).cuda()
    eng = engine()
    assert eng.get_option("flow_fused") == 1
    flow1, img1 = eng.optical_flow(frames, want_flow=True, want_image=True)
    eng.set_option("flow_fused", 0)
    try:
        flow0, img0 = eng.optical_flow(frames, want_flow=True, want_image=True)
    finally:
        eng.set_option("flow_fused", 1)
    assert bool(torch.isfinite(flow1).all())
    assert torch.equal(flow1, flow0), (float((flow1 - flow0).abs().max()), float((flow1 != flow0).float().mean()))
    assert torch.equal(img1, img0)


@pytest.mark.parametrize("h,w,t", [(256, 256, 1), (264, 520, 2), (512, 272, 1), (1080, 1920, 1), (720, 1280, 2), (2160, 3840, 1)])
def test_fused_pyramid_equals_the_per_level_kernels_bit_for_bit(h, w, t):
    """pyramid_fused (the four level inputs of a frame in one pass over its uint8 bytes; frames whose sides are multiples of 8)
    against flow_gray + the per-level blur / resize kernels (option flow_pyramid_fused = 0): same products in the same order, so
    the flow and its image are identical bit for bit - bands that end inside the last 256 columns, segments that end inside the
    last 256 rows, reflected borders on all four sides."""
    frames = torch.from_numpy(np.stack([np.stack(_smooth_pair(h, w, 60 + i)) for i in range(t)])).cuda()
    eng = engine()
    assert eng.get_option("flow_pyramid_fused") == 1
    flow1, img1 = eng.optical_flow(frames, want_flow=True, want_image=True)
    eng.set_option("flow_pyramid_fused", 0)
    try:
        flow0, img0 = eng.optical_flow(frames, want_flow=True, want_image=True)
    finally:
        eng.set_option("flow_pyramid_fused", 1)
    assert bool(torch.isfinite(flow1).all())
    assert torch.equal(flow1, flow0), (float((flow1 - flow0).abs().max()), float((flow1 != flow0).float().mean()))
    assert torch.equal(img1, img0)


def test_flow_does_not_depend_on_the_row_segmentation():
    """A block of the iteration kernels restarts the running column sums (double) at its first row; those sums are exact for these
    magnitudes, so where the segments start must not change a bit of the flow."""
    frames = torch.from_numpy(np.stack([np.stack(_smooth_pair(540, 960, 90))])).cuda()
    eng = engine()
    ref, _ = eng.optical_flow(frames, want_flow=True, want_image=False)
    try:
        for seg in (30, 45, 135, 270):
            eng.set_option("flow_seg_rows", seg)
            got, _ = eng.optical_flow(frames, want_flow=True, want_image=False)
            assert torch.equal(got, ref), seg
    finally:
        eng.set_option("flow_seg_rows", 0)
