"""Sampling semantics / file formats: pure host logic."""
import os

import numpy as np
import pytest
import scipy.io

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import sampling


def test_frame_interval_rule():
    assert sampling.frame_interval(24) == 12 and sampling.frame_interval(29.97002997) == 14
    assert sampling.frame_interval(25) == 12 and sampling.frame_interval(1.5) == 1 and sampling.frame_interval(60) == 30


def test_select_filters_and_zip_truncation():
    s, f, pairs = sampling.sampled_frame_indices(25, 12)        # n%12==0 -> 0,12,24 ; (n-1)%12==0 -> 1,13
    assert s == [0, 12, 24] and f == [1, 13] and pairs == [(0, 1), (12, 13)]
    s, f, pairs = sampling.sampled_frame_indices(26, 12)
    assert pairs == [(0, 1), (12, 13), (24, 25)]
    frames = np.arange(26 * 2 * 2 * 3, dtype=np.uint8).reshape(26, 2, 2, 3)
    p = sampling.pair_frames(frames, 24)
    assert p.shape == (3, 2, 2, 2, 3) and np.array_equal(p[1, 1], frames[13])


def test_file_names_and_mat_roundtrip(tmp_path):
    assert sampling.feature_file_name(0, "resnet50") == "video_1_resnet50_feature_map_original.npy"
    assert sampling.feature_file_name(4, "vit", "360P") == "video_5_vit_feature_map_original_360P.npy"
    g = np.random.default_rng(0)
    paths = [sampling.save_clip_features(str(tmp_path), i, "resnet50", g.standard_normal((3 + i, 7))) for i in range(2)]
    again = sampling.save_clip_features(str(tmp_path), 0, "resnet50", np.zeros((1, 7)), skip_existing=True)
    assert again == paths[0] and np.load(again).shape == (3, 7)          # resume: not overwritten
    mat = sampling.features_matrix(paths)
    assert mat.shape == (2, 7) and np.allclose(mat[1], np.load(paths[1]).mean(axis=0))
    out = sampling.save_mat(str(tmp_path / "f" / "x.mat"), "konvid_1k", mat)
    assert np.allclose(scipy.io.loadmat(out)["konvid_1k"], mat)


def test_frame_files_are_paired_and_read_like_the_reference(tmp_path):
    """`{video}_{n}.png` / `{video}_{n}_next.png` as the reference's ffmpeg step leaves them (src/video_frames_extract.py:51-69): paired
    by NUMERIC index (10 after 2), the list without a partner truncated (src/main_fragment_layerstack.py:283-293), read as cv2.imread
    reads them - BGR, gray frames replicated, alpha dropped."""
    from PIL import Image
    from relax_vqa_amd import sampling
    g = np.random.default_rng(3)
    d = tmp_path / "frames"
    d.mkdir()
    rgb = {n: g.integers(0, 256, (24, 40, 3), dtype=np.uint8) for n in (1, 2, 10, 11)}
    nxt = {n: g.integers(0, 256, (24, 40, 3), dtype=np.uint8) for n in (1, 2, 10)}       # frame 11 has no `next`
    for n, a in rgb.items():
        Image.fromarray(a).save(d / f"my_video_{n}.png")
    for n, a in nxt.items():
        Image.fromarray(a).save(d / f"my_video_{n}_next.png")
    Image.fromarray(rgb[1]).save(d / "other_video_1.png")                                  # another video in the same directory
    pairs = sampling.frame_pair_paths(str(d), "my_video")
    assert [os.path.basename(a) for a, _ in pairs] == ["my_video_1.png", "my_video_2.png", "my_video_10.png"]
    assert [os.path.basename(b) for _, b in pairs] == ["my_video_1_next.png", "my_video_2_next.png", "my_video_10_next.png"]
    clip = sampling.load_clip_from_frames(str(d), "my_video")
    assert clip.shape == (3, 2, 24, 40, 3) and clip.dtype == np.uint8
    for t, n in enumerate((1, 2, 10)):
        assert np.array_equal(clip[t, 0], rgb[n][..., ::-1]) and np.array_equal(clip[t, 1], nxt[n][..., ::-1])
    # gray and RGBA files come out as 3-channel BGR
    Image.fromarray(rgb[1][..., 0]).save(d / "gray_1.png")
    Image.fromarray(np.dstack([rgb[2], np.full((24, 40), 7, np.uint8)])).save(d / "gray_1_next.png")
    c2 = sampling.load_clip_from_frames(str(d), "gray")
    assert np.array_equal(c2[0, 0], np.repeat(rgb[1][..., :1], 3, axis=2)) and np.array_equal(c2[0, 1], rgb[2][..., ::-1])
    with pytest.raises(FileNotFoundError):
        sampling.load_clip_from_frames(str(d), "missing_video")


def test_dataset_pass_from_frame_directories(tmp_path):
    """The reference's data flow end to end on the host side: frame files on disk -> loader threads decode them -> batches (a stand-in
    engine here); an undecodable video costs its own row."""
    import torch
    from PIL import Image
    from relax_vqa_amd import dataset, sampling
    from tests.test_dataset_cpu import F, FakeEngine
    dataset.feature_dim = lambda engine, resnet=True, vit=True, full=False: F
    g = np.random.default_rng(5)
    clips = []
    for v in range(4):
        t = 1 + v % 3
        c = g.integers(0, 255, (t, 2, 32, 48, 3), dtype=np.uint8)
        clips.append(c)
        d = tmp_path / f"v{v}"
        d.mkdir()
        if v == 2:
            (d / "vid_1.png").write_bytes(b"not a png")
            (d / "vid_1_next.png").write_bytes(b"not a png")
            continue
        for k in range(t):
            Image.fromarray(c[k, 0][..., ::-1]).save(d / f"vid_{k * 15}.png")
            Image.fromarray(c[k, 1][..., ::-1]).save(d / f"vid_{k * 15}_next.png")
    matrix, errors = dataset.extract_dataset_clips(lambda i: sampling.load_clip_from_frames(str(tmp_path / f"v{i}"), "vid"), 4, FakeEngine(),
                                                   clips_per_step=2, rank=0, world=1, workers=3)
    assert [i for i, _ in errors] == [2] and bool(torch.isnan(matrix[2]).all())
    for v in (0, 1, 3):
        assert np.array_equal(matrix[v].numpy(), FakeEngine._rows(clips[v]).mean(dim=0).numpy())
