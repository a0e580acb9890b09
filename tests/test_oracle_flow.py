"""Farneback + flow_to_rgb restatement against the reference's own OpenCV output (the shipped 960x540 example pair).
Tolerance pin: OpenCV is absent here, and float rounding in its SIMD kernels differs from numpy's."""
import os

import numpy as np
from PIL import Image

from oracle import flow_ref, fragment_ref


def _load(golden_dir, suffix):
    p = os.path.join(golden_dir, "png_5636101558_3", f"5636101558_3{suffix}.png")
    return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1])


def test_flow_image_matches_reference_png(golden_dir):
    orig, nxt, want = _load(golden_dir, ""), _load(golden_dir, "_next"), _load(golden_dir, "_residual_of")
    flow = flow_ref.farneback(flow_ref.bgr2gray(orig), flow_ref.bgr2gray(nxt))
    assert flow.shape == (540, 960, 2) and flow.dtype == np.float32
    got = flow_ref.flow_to_rgb(flow)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert (diff == 0).mean() > 0.999, (diff == 0).mean()          # measured 0.99976
    assert (diff <= 1).mean() > 0.9999
    # the fragment cut from it: same index map (>= 195 of 196 positions) as the reference's flow fragment
    wf, wp = fragment_ref.extract_important_patches(want, fragment_ref.get_patch_diff(want))
    gf, gp = fragment_ref.extract_important_patches(got, fragment_ref.get_patch_diff(got))
    same = len(set(map(tuple, wp.tolist())) & set(map(tuple, gp.tolist())))
    assert same >= 195, same
    assert np.array_equal(wf, _load(golden_dir, "_residual_of_imp"))


def test_gray_is_opencv_fixed_point():
    img = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], dtype=np.uint8)   # BGR
    assert flow_ref.bgr2gray(img).tolist() == [[29, 150, 76, 128]]


def test_hsv_to_bgr_primaries():
    hsv = np.array([[[0, 255, 255], [60, 255, 255], [120, 255, 255], [30, 255, 200], [0, 0, 77]]], dtype=np.uint8)
    bgr = flow_ref.hsv_to_bgr_u8(hsv)
    assert bgr[0, 0].tolist() == [0, 0, 255] and bgr[0, 1].tolist() == [0, 255, 0] and bgr[0, 2].tolist() == [255, 0, 0]
    assert bgr[0, 3].tolist() == [0, 200, 200] and bgr[0, 4].tolist() == [77, 77, 77]
