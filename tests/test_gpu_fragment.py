"""Stage A parity on the GPU: bit-exact against the oracle and the reference-derived golden vectors."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import fragment_ref
from tests.gpu_common import engine, synth

pytestmark = pytest.mark.gpu


def _run_pairs(frames_np, **kw):
    out = engine().fragment_pairs(torch.from_numpy(frames_np).cuda(), want_scores=True, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _check_pair(out, t, orig, nxt, top_n=196):
    ref = fragment_ref.fragment_pair(orig, nxt, top_n=top_n)
    assert np.array_equal(out["scores"][t].astype(np.float64), ref["score"]), "patch scores differ"
    n = len(ref["positions"])
    assert out["counts"][t] == n
    assert np.array_equal(out["positions"][t, :n], ref["positions"]), "fragment index map differs"
    assert (out["positions"][t, n:] == -1).all()
    assert np.array_equal(out["diff_frag"][t], ref["diff_frag"]), "residual fragment differs"
    assert np.array_equal(out["ori_frag"][t], ref["ori_frag"]), "original fragment differs"


def test_golden_synthetic_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "fragment_synthetic.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        orig, nxt = z[f"{name}/orig"], z[f"{name}/next"]
        out = _run_pairs(np.stack([orig, nxt])[None])
        assert np.array_equal(out["scores"][0].astype(np.float64), z[f"{name}/score"]), name
        n = len(z[f"{name}/positions"])
        assert out["counts"][0] == n and np.array_equal(out["positions"][0, :n], z[f"{name}/positions"]), name
        assert np.array_equal(out["diff_frag"][0], z[f"{name}/diff_frag"]), name
        assert np.array_equal(out["ori_frag"][0], z[f"{name}/ori_frag"]), name


def test_real_video_known_answer(golden_dir):
    d = os.path.join(golden_dir, "png_5636101558_3")

    def load(suffix):
        return np.ascontiguousarray(np.asarray(Image.open(os.path.join(d, f"5636101558_3{suffix}.png")).convert("RGB"))[..., ::-1])

    orig, nxt = load(""), load("_next")
    out = _run_pairs(np.stack([orig, nxt])[None])
    assert np.array_equal(out["diff_frag"][0], load("_residual_imp"))
    assert np.array_equal(out["ori_frag"][0], load("_ori_frag"))
    flow = load("_residual_of")
    fo = engine().fragment_image(torch.from_numpy(flow[None]).cuda())
    assert np.array_equal(fo["frag"][0].cpu().numpy(), load("_residual_of_imp"))
    merged = engine().merge_fragments(torch.from_numpy(out["diff_frag"][:1]).cuda(), fo["frag"])
    assert np.array_equal(merged[0].cpu().numpy(), load("_residual_merged_frag"))


@pytest.mark.parametrize("h,w", [(540, 960), (720, 1280), (250, 333), (100, 130), (16, 16), (15, 40), (224, 224)])
def test_sizes_aligned_and_ragged(h, w):
    clip = synth.synthetic_clip(3, h, w, clip_id=h + w)
    out = _run_pairs(clip)
    for t in range(3):
        _check_pair(out, t, clip[t, 0], clip[t, 1])


def test_full_size_1080p_clip():
    clip = synth.synthetic_clip(4, 1080, 1920, clip_id=9)
    out = _run_pairs(clip)
    for t in range(4):
        _check_pair(out, t, clip[t, 0], clip[t, 1])
    # size-independent properties: positions strictly increasing in raster order, every selected score >= every
    # unselected score, checksum of the fragments equals checksum of the gathered patches
    for t in range(4):
        pos = out["positions"][t]
        flat = pos[:, 0] * (1920 // 16) + pos[:, 1]
        assert (np.diff(flat) > 0).all()
        s = out["scores"][t].ravel()
        sel = np.zeros(s.size, bool)
        sel[flat] = True
        assert s[sel].min() >= s[~sel].max()
        assert out["diff_frag"][t].astype(np.uint64).sum() == s[sel].sum()


def test_2160p_pair():
    clip = synth.synthetic_clip(1, 2160, 3840, clip_id=21)
    out = _run_pairs(clip)
    _check_pair(out, 0, clip[0, 0], clip[0, 1])


def test_ties_resolved_by_lowest_index():
    h, w = 320, 480
    zeros = np.zeros((1, 2, h, w, 3), np.uint8)          # every score 0 -> first 196 patches in raster order
    out = _run_pairs(zeros)
    _check_pair(out, 0, zeros[0, 0], zeros[0, 1])
    a = np.zeros((h, w, 3), np.uint8)
    b = a.copy()
    b[:, :, 0] = 7                                         # every patch scores 16*16*7: all tied, non-zero
    out = _run_pairs(np.stack([a, b])[None])
    _check_pair(out, 0, a, b)
    g = np.random.default_rng(5)
    lvl = g.integers(0, 3, (h // 16, w // 16)).astype(np.uint8)   # three score levels, ties straddle rank 196
    b = np.repeat(np.repeat(lvl, 16, 0), 16, 1)[..., None].repeat(3, 2)
    out = _run_pairs(np.stack([a, b])[None])
    _check_pair(out, 0, a, b)


@pytest.mark.parametrize("top_n", [0, 1, 50, 195])
def test_top_n_parameter(top_n):
    clip = synth.synthetic_clip(1, 240, 320, clip_id=77)
    out = _run_pairs(clip, top_n=top_n)
    _check_pair(out, 0, clip[0, 0], clip[0, 1], top_n=top_n)


def test_merge_matches_oracle_on_all_byte_pairs():
    a = np.repeat(np.arange(256, dtype=np.uint8), 256)
    b = np.tile(np.arange(256, dtype=np.uint8), 256)
    got = engine().merge_fragments(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert np.array_equal(got, fragment_ref.merge_fragments(a, b))
    got = engine().merge_fragments(torch.from_numpy(a[:1001]).cuda(), torch.from_numpy(b[:1001]).cuda()).cpu().numpy()
    assert np.array_equal(got, fragment_ref.merge_fragments(a[:1001], b[:1001]))


def test_gather_with_given_positions():
    clip = synth.synthetic_clip(2, 272, 400, clip_id=31)
    out = _run_pairs(clip)
    nxt = torch.from_numpy(np.ascontiguousarray(clip[:, 1])).cuda()
    got = engine().gather_patches(nxt, torch.from_numpy(out["positions"]), torch.from_numpy(out["counts"])).cpu().numpy()
    for t in range(2):
        n = out["counts"][t]
        assert np.array_equal(got[t], fragment_ref.get_original_frame_patches(clip[t, 1], out["positions"][t, :n]))


def test_gather_rejects_out_of_range_positions():
    img = np.random.default_rng(2).integers(1, 256, (1, 64, 80, 3), dtype=np.uint8)
    pos = torch.full((1, 196, 2), -1, dtype=torch.int32)
    pos[0, 0] = torch.tensor([1, 2])
    pos[0, 1] = torch.tensor([400, 3])          # far outside the 4 x 5 patch grid
    pos[0, 2] = torch.tensor([-7, 0])
    got = engine().gather_patches(torch.from_numpy(img).cuda(), pos, torch.tensor([3], dtype=torch.int32)).cpu().numpy()
    assert np.array_equal(got[0, :16, :16], img[0, 16:32, 32:48])
    assert not got[0, :16, 16:48].any()          # the two bad tiles are zero, nothing was read out of bounds
