"""Stage A on the GPU against the oracle on RANDOM shapes and contents (hypothesis): the integer path must be bit-exact for every
frame size (ragged, smaller than a patch, fewer than 196 patches), every top_n, and for contents that tie (constant frames, a few
score levels), saturate (0 / 255) or barely move.  Sizes are kept small: the oracle is the checker, not the thing timed.
The examples are derandomized (the same ones on every box).  First catch: top_n = 0 on a frame with patches left the selection
threshold undefined (select_topn searched a histogram bin for "remaining = 0"); fixed in csrc/fragment.hip."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import fragment_ref
from tests.gpu_common import engine

pytestmark = pytest.mark.gpu


def _content(kind, g, t, h, w):
    if kind == "random":
        a = g.integers(0, 256, (t, h, w, 3), dtype=np.uint8)
        b = g.integers(0, 256, (t, h, w, 3), dtype=np.uint8)
    elif kind == "constant":                       # every patch ties
        a = np.full((t, h, w, 3), int(g.integers(0, 256)), np.uint8)
        b = np.full((t, h, w, 3), int(g.integers(0, 256)), np.uint8)
    elif kind == "levels":                         # a handful of score levels: ties straddle the cut
        a = np.zeros((t, h, w, 3), np.uint8)
        lvl = g.integers(0, 4, (t, (h + 15) // 16, (w + 15) // 16)).astype(np.uint8)
        b = np.repeat(np.repeat(lvl, 16, 1), 16, 2)[:, :h, :w, None].repeat(3, 3)
    elif kind == "saturated":                      # |a - b| = 255 in places: the largest patch sums (195840)
        a = (g.integers(0, 2, (t, h, w, 3)) * 255).astype(np.uint8)
        b = (g.integers(0, 2, (t, h, w, 3)) * 255).astype(np.uint8)
    else:                                          # sparse motion: most patches score 0
        a = g.integers(0, 256, (t, h, w, 3), dtype=np.uint8)
        b = a.copy()
        for _ in range(int(g.integers(0, 6))):
            y, x = int(g.integers(0, h)), int(g.integers(0, w))
            b[:, y:y + int(g.integers(1, 40)), x:x + int(g.integers(1, 40))] ^= np.uint8(g.integers(1, 256))
    return np.ascontiguousarray(np.stack([a, b], axis=1))      # [T, 2, H, W, 3]


@settings(max_examples=200, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(h=st.integers(1, 150), w=st.integers(1, 210), t=st.integers(1, 3), top_n=st.sampled_from([0, 1, 7, 50, 195, 196]),
       kind=st.sampled_from(["random", "constant", "levels", "saturated", "sparse"]), seed=st.integers(0, 2 ** 31 - 1))
def test_fragment_pairs_bit_exact_on_random_cases(h, w, t, top_n, kind, seed):
    frames = _content(kind, np.random.default_rng(seed), t, h, w)
    out = engine().fragment_pairs(torch.from_numpy(frames).cuda(), top_n=top_n, want_scores=True)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(t):
        ref = fragment_ref.fragment_pair(frames[i, 0], frames[i, 1], top_n=top_n)
        assert np.array_equal(out["scores"][i].astype(np.float64), ref["score"]), "patch scores differ"
        n = len(ref["positions"])
        assert out["counts"][i] == n
        assert np.array_equal(out["positions"][i, :n], ref["positions"]), "fragment index map differs"
        assert (out["positions"][i, n:] == -1).all()
        assert np.array_equal(out["diff_frag"][i], ref["diff_frag"]), "residual fragment differs"
        assert np.array_equal(out["ori_frag"][i], ref["ori_frag"]), "original fragment differs"


@settings(max_examples=120, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(h=st.integers(1, 150), w=st.integers(1, 210), n=st.integers(1, 3), top_n=st.sampled_from([0, 3, 196]),
       kind=st.sampled_from(["random", "constant", "levels", "sparse"]), seed=st.integers(0, 2 ** 31 - 1))
def test_fragment_image_and_merge_bit_exact_on_random_cases(h, w, n, top_n, kind, seed):
    """The single-image form (flow image -> fragment) and the 50/50 merge on the same random material."""
    g = np.random.default_rng(seed)
    imgs = _content(kind, g, n, h, w)[:, 1]
    out = engine().fragment_image(torch.from_numpy(np.ascontiguousarray(imgs)).cuda(), top_n=top_n, want_scores=True)
    torch.cuda.synchronize()
    for i in range(n):
        diff = fragment_ref.get_patch_diff(imgs[i])
        ref_frag, ref_pos = fragment_ref.extract_important_patches(imgs[i], diff, top_n=top_n)
        k = len(ref_pos)
        assert np.array_equal(out["scores"][i].cpu().numpy().astype(np.float64), diff)
        assert int(out["counts"][i]) == k
        assert np.array_equal(out["positions"][i, :k].cpu().numpy(), ref_pos)
        assert np.array_equal(out["frag"][i].cpu().numpy(), ref_frag)
    other = torch.from_numpy(g.integers(0, 256, tuple(out["frag"].shape), dtype=np.uint8)).cuda()
    merged = engine().merge_fragments(out["frag"], other).cpu().numpy()
    assert np.array_equal(merged, fragment_ref.merge_fragments(out["frag"].cpu().numpy(), other.cpu().numpy()))
