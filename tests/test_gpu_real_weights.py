"""Real pretrained checkpoints through the engine (round-4 review, item 8).  The reference always runs pretrained weights -
`models.resnet50(pretrained=True)` (src/extractor/visualise_resnet.py:21), the DINO ViT-B/16 hub file
(src/extractor/visualise_vit_layer.py:304-329) - and neither is in the image (no network).  These tests ARM THEMSELVES when
RELAX_RESNET50_WEIGHTS / RELAX_VIT_WEIGHTS name a state-dict file: they load it through the product loader (runtime._load_file: wrappers and
key prefixes as the hub files have them), run the 32-fragment parity of tests/test_gpu_bench_path.py with THOSE weights against the oracle,
under every fp32-grade arithmetic, and print the measured error (DINO's outlier channels and real BatchNorm statistics are value ranges the
synthetic sets do not have; the f16x2 scales are bounds, so they hold for any weights - this is where that is exercised).  Skipped with the
reason printed otherwise."""
import os

import numpy as np
import pytest
import torch

from oracle import fragment_ref, pooling_ref, resnet50_ref, vit_ref
from relax_vqa_amd import runtime
from tests.gpu_common import _weights, assert_close, engine, synth

pytestmark = pytest.mark.gpu
N_FRAGS = 32


def _fragments():
    frs = []
    for i in range(N_FRAGS):
        o, nx = synth.synthetic_pair(240, 320, 9000 + i)
        f = fragment_ref.fragment_pair(o, nx)
        frs.append(f["ori_frag"] if i % 2 == 0 else f["diff_frag"])
    return np.stack(frs)


def _checkpoint(env_var, what):
    path = os.environ.get(env_var)
    if not path:
        pytest.skip(f"{env_var} is not set: no pretrained {what} checkpoint on this box (the image has no network); the test arms itself "
                    f"when the variable names a state-dict file")
    if not os.path.isfile(path):
        pytest.skip(f"{env_var}={path}: no such file")
    return runtime._load_file(path)


def test_real_resnet50_checkpoint_32_fragments_against_the_oracle(each_precision):
    sd = _checkpoint("RELAX_RESNET50_WEIGHTS", "ResNet-50 (torchvision IMAGENET1K_V1)")
    eng = engine()
    eng.load_resnet50(sd)
    _weights["rn_loaded"] = "real"                     # (the shared engine no longer holds a synthetic set)
    frags = _fragments()
    tsd = resnet50_ref.to_torch_state_dict(sd)
    want_taps, _ = resnet50_ref.forward_taps(tsd, resnet50_ref.preprocess_bgr_u8(frags))
    ls, pool, taps = eng.resnet50_features(torch.from_numpy(frags).cuda(), taps=range(15))
    worst = 0.0
    for i, name in enumerate(pooling_ref.RESNET50_TAPS):
        w = want_taps[name].numpy()
        worst = max(worst, float(np.abs(taps[i].cpu().numpy() - w).max() / np.abs(w).max()))
        assert_close(taps[i], w, f"real weights, {name}")
    assert_close(ls, resnet50_ref.layer_stack_features(tsd, frags), "real weights, layer stack")
    assert_close(pool, resnet50_ref.pool_features(tsd, frags), "real weights, pool")
    print(f"\nreal ResNet-50 checkpoint ({each_precision}): worst tap max-rel error {worst:.3e}")


def test_real_vit_checkpoint_32_fragments_against_the_oracle(each_precision):
    sd = _checkpoint("RELAX_VIT_WEIGHTS", "DINO ViT-B/16")
    eng = engine()
    eng.load_vit(sd, "vit_base")
    _weights["vit_loaded"] = "real"
    frags = _fragments()
    tsd = vit_ref.to_torch_state_dict(sd)
    want = vit_ref.tokens(tsd, frags, 12)
    tokens, pooled = eng.vit_features(torch.from_numpy(frags).cuda(), tokens=True, pooled=True)
    assert torch.isfinite(tokens).all(), "non-finite tokens: a static f16x2 scale was exceeded"
    assert_close(tokens, want, "real weights, ViT tokens")
    assert_close(pooled, vit_ref.pool_features(tsd, frags, 12), "real weights, ViT pooled")
    err = float(np.abs(tokens.cpu().numpy() - want).max() / np.abs(want).max())
    print(f"\nreal DINO ViT-B/16 checkpoint ({each_precision}): tokens max-rel error {err:.3e}")
