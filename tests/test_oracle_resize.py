"""Pin the resize restatement against Pillow itself (the reference's dependency for this step)."""
import numpy as np
import pytest
from PIL import Image

from oracle import resize_ref


@pytest.mark.parametrize("h,w", [(540, 960), (270, 480), (224, 300), (300, 224), (100, 130), (1080, 1920)])
@pytest.mark.parametrize("filt,pil", [(resize_ref.BILINEAR, Image.BILINEAR), (resize_ref.LANCZOS, Image.LANCZOS)])
def test_matches_pillow_bit_exactly(h, w, filt, pil):
    if h * w > 600 * 1000 and filt == resize_ref.LANCZOS and False:
        pytest.skip("slow")
    g = np.random.default_rng(h * 7 + w)
    img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[: h // 3] = (img[: h // 3].astype(np.int32) // 8 * 8).astype(np.uint8)   # some structure, not only noise
    want = np.asarray(Image.fromarray(img).resize((224, 224), pil))
    got = resize_ref.resize(img, 224, 224, filt)
    assert np.array_equal(got, want)


def test_identity_size_is_a_copy():
    img = np.random.default_rng(1).integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(resize_ref.resize(img, 224, 224, resize_ref.LANCZOS), img)
