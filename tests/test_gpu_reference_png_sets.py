"""The reference's own example PNG sets at the headline (1080p) and config-5 (2160p) resolutions under the HIP path.

These are real video frames and the images the reference derived from them with OpenCV (src/main_fragment_layerstack.py:302-325,
visualisation/visualisation_example/original_TelevisionClip_1080P-68c6, original_Sports_2160P-0455; copied as data by
oracle/make_golden.py).  Integer work is bit-exact; the flow image is held to the bars of the 540p set (tests/test_gpu_flow.py)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import fragment_ref
from tests.gpu_common import engine

pytestmark = pytest.mark.gpu

TV = "TelevisionClip_1080P-68c6_1"
SPORTS = "Sports_2160P-0455_1"


def _load(golden_dir, stem, suffix):
    p = os.path.join(golden_dir, "png_" + stem, f"{stem}{suffix}.png")
    return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1])    # PIL gives RGB, cv2 holds BGR


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_1080p_pair_fragments_are_the_references(golden_dir):
    """absdiff + patch scores + top-196 + both gathers on the reference's 1080p pair: its `_residual_imp.png` and `_ori_frag.png`
    byte for byte, the index map equal to the pinned restatement's (and no tie across rank 196 / 197 in this set)."""
    orig, nxt = _load(golden_dir, TV, ""), _load(golden_dir, TV, "_next")
    assert orig.shape == (1080, 1920, 3)
    out = engine().fragment_pairs(_dev(np.stack([orig, nxt])[None]), want_scores=True)
    ref = fragment_ref.fragment_pair(orig, nxt)
    s = np.sort(ref["score"].ravel())[::-1]
    assert s[195] != s[196]
    assert np.array_equal(out["scores"][0].cpu().numpy().astype(np.float64), ref["score"])
    assert int(out["counts"][0]) == 196
    assert np.array_equal(out["positions"][0].cpu().numpy(), ref["positions"])
    assert np.array_equal(out["diff_frag"][0].cpu().numpy(), _load(golden_dir, TV, "_residual_imp"))
    assert np.array_equal(out["ori_frag"][0].cpu().numpy(), _load(golden_dir, TV, "_ori_frag"))


def test_1080p_flow_fragment_and_merge_from_the_references_flow_image(golden_dir):
    """process_patches on OpenCV's flow image and merge_fragments (src/main_fragment_layerstack.py:319-325): bit-exact."""
    flow_img = _load(golden_dir, TV, "_residual_of")
    fo = engine().fragment_image(_dev(flow_img[None]))
    assert np.array_equal(fo["frag"][0].cpu().numpy(), _load(golden_dir, TV, "_residual_of_imp"))
    merged = engine().merge_fragments(_dev(_load(golden_dir, TV, "_residual_imp")[None]), fo["frag"])
    assert np.array_equal(merged[0].cpu().numpy(), _load(golden_dir, TV, "_residual_merged_frag"))


def test_1080p_flow_image_against_opencvs(golden_dir):
    """Farneback + flow_to_rgb on the 1080p pair against the reference's `_residual_of.png` (OpenCV 4.9), at the bars of the 540p
    set: >= 99.8 % of the bytes identical, >= 99.99 % within 1, >= 195 of 196 flow-fragment positions, and the merged fragment."""
    orig, nxt, want = _load(golden_dir, TV, ""), _load(golden_dir, TV, "_next"), _load(golden_dir, TV, "_residual_of")
    frames = _dev(np.stack([orig, nxt])[None])
    _, img = engine().optical_flow(frames)
    img_np = img[0].cpu().numpy()
    d = np.abs(img_np.astype(np.int32) - want.astype(np.int32))
    assert (d == 0).mean() > 0.998 and (d <= 1).mean() > 0.9999, ((d == 0).mean(), (d <= 1).mean(), d.max())
    fo = engine().fragment_image(img)
    n = int(fo["counts"][0])
    got = set(map(tuple, fo["positions"][0, :n].cpu().numpy().tolist()))
    _, wp = fragment_ref.extract_important_patches(want, fragment_ref.get_patch_diff(want))
    assert len(got & set(map(tuple, wp.tolist()))) >= 195
    fr = engine().fragment_pairs(frames)
    merged = engine().merge_fragments(fr["diff_frag"], fo["frag"])[0].cpu().numpy()
    assert (merged == _load(golden_dir, TV, "_residual_merged_frag")).mean() > 0.995


def test_2160p_flow_fragment_and_merge(golden_dir):
    """The 2160p set holds OpenCV's flow image and the fragments only: patch scores / top-196 / gather over 135 x 240 patches and
    the merge with the reference's residual fragment, bit-exact."""
    flow_img = _load(golden_dir, SPORTS, "_residual_of")
    assert flow_img.shape == (2160, 3840, 3)
    fo = engine().fragment_image(_dev(flow_img[None]), want_scores=True)
    assert np.array_equal(fo["scores"][0].cpu().numpy().astype(np.float64), fragment_ref.get_patch_diff(flow_img))
    assert np.array_equal(fo["frag"][0].cpu().numpy(), _load(golden_dir, SPORTS, "_residual_of_imp"))
    merged = engine().merge_fragments(_dev(_load(golden_dir, SPORTS, "_residual_imp")[None]), fo["frag"])
    assert np.array_equal(merged[0].cpu().numpy(), _load(golden_dir, SPORTS, "_residual_merged_frag"))
