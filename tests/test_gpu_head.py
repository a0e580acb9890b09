"""Quality head on the GPU against the golden vector made with the reference's Mlp class and its
real KoNViD scaler pickles, and against the oracle on a bigger batch."""
import os

import numpy as np
import pytest
import torch

from oracle import mlp_ref
from tests.gpu_common import engine, synth

pytestmark = pytest.mark.gpu


def test_head_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "mlp_head.npz"))
    sd = synth.mlp_head_state_dict(35203, 256, seed=23)
    eng = engine()
    eng.load_mlp_head({("module." + k): v for k, v in sd.items()}, z["scale"], z["min"], z["imputer_statistics"])
    got = eng.mlp_head(torch.from_numpy(z["features"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, z["expected"], rtol=1e-3, atol=1e-3)   # north_star tolerance; measured ~1e-6
    assert np.abs(got - z["expected"]).max() / np.abs(z["expected"]).max() < 1e-5


def test_head_batch_against_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, "mlp_head.npz"))
    sd = synth.mlp_head_state_dict(35203, 256, seed=23)
    eng = engine()
    eng.load_mlp_head(sd, z["scale"], z["min"], z["imputer_statistics"])
    g = np.random.default_rng(4)
    feats = np.repeat(z["features"], 50, axis=0) * g.uniform(0.9, 1.1, (150, 1)).astype(np.float32)
    feats[7, 100:120] = np.nan
    want = mlp_ref.predict(sd, feats, z["imputer_statistics"], z["scale"], z["min"])
    got = eng.mlp_head(torch.from_numpy(feats).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-3)


def test_end_to_end_demo(golden_dir):
    """frames -> 35203-d vector -> imputer/scaler -> MLP -> score, against the oracle pipeline."""
    from oracle import fragment_ref, resize_ref, resnet50_ref, vit_ref
    from relax_vqa_amd import demo_test, runtime
    z = np.load(os.path.join(golden_dir, "mlp_head.npz"))
    rn, vit = synth.resnet50_state_dict(), synth.vit_state_dict("vit_base")
    head = synth.mlp_head_state_dict(35203, 256, seed=23)
    runtime.set_weights(resnet50=rn, vit=vit, vit_name="vit_base")
    demo_test.load_head(head, z["imputer_statistics"], (z["scale"], z["min"]))
    T = 2
    clip = synth.synthetic_clip(T, 272, 400, clip_id=41)
    got = demo_test.evaluate_video_quality(clip, "konvid_1k", flow=False)
    # oracle: same vector assembled on the CPU
    tr, tv = resnet50_ref.to_torch_state_dict(rn), vit_ref.to_torch_state_dict(vit)
    refs = [fragment_ref.fragment_pair(clip[t, 0], clip[t, 1]) for t in range(T)]
    ori, res = np.stack([r["ori_frag"] for r in refs]), np.stack([r["diff_frag"] for r in refs])
    whole_b = np.stack([resize_ref.resize(clip[t, 0], 224, 224, resize_ref.BILINEAR) for t in range(T)])
    whole_l = np.stack([resize_ref.resize(clip[t, 0], 224, 224, resize_ref.LANCZOS) for t in range(T)])
    vec = np.concatenate([
        resnet50_ref.layer_stack_features(tr, whole_b).mean(0), vit_ref.pool_features(tv, whole_l, 12).mean(0),
        np.concatenate([resnet50_ref.layer_stack_features(tr, ori), resnet50_ref.pool_features(tr, res)], 1).mean(0),
        np.concatenate([vit_ref.pool_features(tv, ori, 12), vit_ref.pool_features(tv, res, 12)], 1).mean(0)])
    assert vec.shape == (35203,)
    want = mlp_ref.rescale_0_100_to_1_5(mlp_ref.predict(head, vec[None], z["imputer_statistics"], z["scale"], z["min"])[0])
    assert abs(got - want) <= 1e-3 * abs(want), (got, want)
