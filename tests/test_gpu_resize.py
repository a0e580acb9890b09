"""Whole-frame resize front-end on the GPU: bit-exact against Pillow and the oracle."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import resize_ref
from tests.gpu_common import engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w", [(540, 960), (1080, 1920), (720, 1280), (270, 480), (224, 224), (224, 300), (301, 224),
                                 (100, 130), (250, 333), (2160, 3840)])
def test_matches_pillow(h, w):
    g = np.random.default_rng(h + 3 * w)
    n = 1 if h * w > 2e6 else 3
    frames = g.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[0, : h // 2] = (frames[0, : h // 2] // 16 * 16)
    bil, lan = engine().resize_frames(torch.from_numpy(frames).cuda())
    bil, lan = bil.cpu().numpy(), lan.cpu().numpy()
    for i in range(n):
        img = Image.fromarray(frames[i])
        assert np.array_equal(bil[i], np.asarray(img.resize((224, 224), Image.BILINEAR))), f"bilinear frame {i}"
        assert np.array_equal(lan[i], np.asarray(img.resize((224, 224), Image.LANCZOS))), f"lanczos frame {i}"


def test_matches_oracle_and_single_filter_requests():
    frames = np.random.default_rng(5).integers(0, 256, (2, 360, 640, 3), dtype=np.uint8)
    bil, none = engine().resize_frames(torch.from_numpy(frames).cuda(), bilinear=True, lanczos=False)
    assert none is None
    assert np.array_equal(bil[1].cpu().numpy(), resize_ref.resize(frames[1], 224, 224, resize_ref.BILINEAR))
    none, lan = engine().resize_frames(torch.from_numpy(frames).cuda(), bilinear=False, lanczos=True)
    assert none is None
    assert np.array_equal(lan[0].cpu().numpy(), resize_ref.resize(frames[0], 224, 224, resize_ref.LANCZOS))
