import os

import numpy as np
import pytest

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import synth
from oracle import vit_ref


@pytest.mark.parametrize("adversarial", [False, True])
@pytest.mark.parametrize("name,heads", [("vit_tiny", 3), ("vit_base", 12)])
def test_vit_tokens_match_reference_golden(golden_dir, name, heads, adversarial):
    """Expected tokens come from the reference's own VisionTransformer class (oracle/make_golden.py), for the regular synthetic
    weights and for the adversarial set (attention logits of +-20, LayerNorm gammas of mixed sign)."""
    z = np.load(os.path.join(golden_dir, f"{name}{'_adv' if adversarial else ''}_tokens.npz"))
    sd_np = synth.vit_state_dict(name, 16, seed=11, adversarial=adversarial)
    probe = np.float64([float(np.sum(v.astype(np.float64))) for v in sd_np.values()]).sum()
    assert probe == float(z["weight_probe"]), "synthetic weight generator drifted from the fixture"
    t = vit_ref.tokens(vit_ref.to_torch_state_dict(sd_np), z["frags"], heads)
    # same container / same torch build reproduces bit-exactly; allow fp32 reassociation elsewhere
    np.testing.assert_allclose(t, z["tokens"], rtol=1e-4, atol=1e-4)


def test_vit_pool_dims(golden_dir):
    z = np.load(os.path.join(golden_dir, "vit_tiny_tokens.npz"))
    sd = vit_ref.to_torch_state_dict(synth.vit_state_dict("vit_tiny", 16, seed=11))
    f = vit_ref.pool_features(sd, z["frags"], heads=3)
    assert f.shape == (2, 3 * 192) and f.dtype == np.float32


def test_adversarial_vit_weights_make_peaked_attention(golden_dir):
    """The adversarial set does what it is for: first-block attention logits span tens of units, so softmax rows are near
    one-hot next to flat ones."""
    import torch
    sd = vit_ref.to_torch_state_dict(synth.vit_state_dict("vit_tiny", 16, seed=11, adversarial=True))
    z = np.load(os.path.join(golden_dir, "vit_tiny_adv_tokens.npz"))
    x = vit_ref.preprocess_bgr_u8(z["frags"][:1])
    t = torch.nn.functional.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    t = torch.cat((sd["cls_token"], t), dim=1) + sd["pos_embed"]
    y = torch.nn.functional.layer_norm(t, (192,), sd["blocks.0.norm1.weight"], sd["blocks.0.norm1.bias"], 1e-6)
    qkv = torch.nn.functional.linear(y, sd["blocks.0.attn.qkv.weight"], sd["blocks.0.attn.qkv.bias"]).reshape(1, 197, 3, 3, 64)
    logits = (qkv[:, :, 0].transpose(1, 2) @ qkv[:, :, 1].transpose(1, 2).transpose(-2, -1)) / 8.0
    assert float(logits.max() - logits.min()) > 30.0
    assert float(logits.softmax(dim=-1).max(dim=-1).values.max()) > 0.9
