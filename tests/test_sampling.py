"""Sampling semantics / file formats: pure host logic."""
import numpy as np
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
