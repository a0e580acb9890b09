"""Oracle vs the golden vectors produced from the reference (CPU)."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import fragment_ref


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "fragment_synthetic.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


def test_synthetic_pairs_match_reference(golden_dir):
    for name, c in _cases(golden_dir).items():
        o = fragment_ref.fragment_pair(c["orig"], c["next"])
        assert np.array_equal(o["score"], c["score"]), name
        assert np.array_equal(o["positions"], c["positions"]), name
        assert np.array_equal(o["diff_frag"], c["diff_frag"]), name
        assert np.array_equal(o["ori_frag"], c["ori_frag"]), name


def test_loop_scoring_equals_vectorised(golden_dir):
    c = _cases(golden_dir)["s100x130_ragged"]
    r = fragment_ref.absdiff(c["next"], c["orig"])
    assert np.array_equal(fragment_ref.get_patch_diff_loop(r), fragment_ref.get_patch_diff(r))


def test_fewer_than_196_patches_leaves_canvas_zero(golden_dir):
    c = _cases(golden_dir)["s96x128_few"]
    o = fragment_ref.fragment_pair(c["orig"], c["next"])
    assert len(o["positions"]) == 48
    assert not o["diff_frag"][64:].any() and not o["ori_frag"][64:].any()   # tiles 48.. stay zero


def test_real_video_known_answer(golden_dir):
    """The reference's own example frames (960x540): every derived PNG reproduced bit-exactly."""
    d = os.path.join(golden_dir, "png_5636101558_3")

    def load(suffix):
        return np.ascontiguousarray(np.asarray(Image.open(os.path.join(d, f"5636101558_3{suffix}.png")).convert("RGB"))[..., ::-1])

    orig, nxt = load(""), load("_next")
    assert np.array_equal(fragment_ref.absdiff(nxt, orig), load("_residual"))
    o = fragment_ref.fragment_pair(orig, nxt)
    assert np.array_equal(o["diff_frag"], load("_residual_imp"))
    assert np.array_equal(o["ori_frag"], load("_ori_frag"))
    flow = load("_residual_of")
    ffrag, _ = fragment_ref.extract_important_patches(flow, fragment_ref.get_patch_diff(flow))
    assert np.array_equal(ffrag, load("_residual_of_imp"))
    assert np.array_equal(fragment_ref.merge_fragments(o["diff_frag"], ffrag), load("_residual_merged_frag"))


def _load_set(golden_dir, stem, suffix):
    p = os.path.join(golden_dir, "png_" + stem, f"{stem}{suffix}.png")
    return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1])


def test_reference_1080p_set(golden_dir):
    """The reference's 1080p example pair (the headline resolution): fragments, flow fragment and merge reproduced bit-exactly."""
    stem = "TelevisionClip_1080P-68c6_1"
    orig, nxt = _load_set(golden_dir, stem, ""), _load_set(golden_dir, stem, "_next")
    o = fragment_ref.fragment_pair(orig, nxt)
    assert np.array_equal(o["diff_frag"], _load_set(golden_dir, stem, "_residual_imp"))
    assert np.array_equal(o["ori_frag"], _load_set(golden_dir, stem, "_ori_frag"))
    flow = _load_set(golden_dir, stem, "_residual_of")
    ffrag, _ = fragment_ref.extract_important_patches(flow, fragment_ref.get_patch_diff(flow))
    assert np.array_equal(ffrag, _load_set(golden_dir, stem, "_residual_of_imp"))
    assert np.array_equal(fragment_ref.merge_fragments(o["diff_frag"], ffrag), _load_set(golden_dir, stem, "_residual_merged_frag"))


def test_reference_2160p_set(golden_dir):
    stem = "Sports_2160P-0455_1"
    flow = _load_set(golden_dir, stem, "_residual_of")
    ffrag, _ = fragment_ref.extract_important_patches(flow, fragment_ref.get_patch_diff(flow))
    assert np.array_equal(ffrag, _load_set(golden_dir, stem, "_residual_of_imp"))
    assert np.array_equal(fragment_ref.merge_fragments(_load_set(golden_dir, stem, "_residual_imp"), ffrag),
                          _load_set(golden_dir, stem, "_residual_merged_frag"))


def test_tie_rule_is_lowest_index_first():
    diff = np.zeros((20, 20))
    pos = fragment_ref.select_positions(diff, 196)
    flat = pos[:, 0] * 20 + pos[:, 1]
    assert np.array_equal(flat, np.arange(196))
    diff[19, 19] = 5
    diff[0, 3] = 5
    flat = fragment_ref.select_positions(diff, 3)
    assert flat.tolist() == [[0, 0], [0, 3], [19, 19]]


def test_merge_rounds_half_to_even():
    a = np.array([[[1, 2, 3]]], dtype=np.uint8)
    b = np.array([[[2, 3, 4]]], dtype=np.uint8)
    assert fragment_ref.merge_fragments(a, b).ravel().tolist() == [2, 2, 4]   # 1.5->2, 2.5->2, 3.5->4


@pytest.mark.parametrize("h,w", [(16, 16), (15, 40), (33, 17)])
def test_tiny_and_sub_patch_frames(h, w):
    g = np.random.default_rng(h * 100 + w)
    a = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    b = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    o = fragment_ref.fragment_pair(a, b)
    assert o["score"].shape == (h // 16, w // 16)
    assert len(o["positions"]) == (h // 16) * (w // 16)
