import os

import numpy as np
import pytest

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import synth
from oracle import vit_ref


@pytest.mark.parametrize("adversarial", [False, True, "outliers"], ids=["regular", "adversarial", "outliers"])
@pytest.mark.parametrize("name,heads", [("vit_tiny", 3), ("vit_base", 12)])
def test_vit_tokens_match_reference_golden(golden_dir, name, heads, adversarial):
    """Expected tokens come from the reference's own VisionTransformer class (oracle/make_golden.py), for the regular synthetic
    weights, for the adversarial set (attention logits of +-20, LayerNorm gammas of mixed sign) and for the outlier set (five
    residual-stream channels hundreds of times the median, LayerNorm gains up to 10 on them, a near-one-hot head)."""
    z = np.load(os.path.join(golden_dir, f"{name}{ {False: '', True: '_adv', 'outliers': '_out'}[adversarial] }_tokens.npz"))
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


def test_outlier_vit_weights_have_the_pathology_of_real_checkpoints(golden_dir):
    """The outlier set does what it is for: in the residual stream behind block 6 five channels are hundreds of times the median
    channel, the LayerNorm output that feeds the qkv GEMM is dominated by them (so a Cauchy-Schwarz bound over all 768 channels is
    loose by 2^6 or more for the typical channel), head 0 has near-one-hot softmax rows - and the network stays well conditioned
    (fp32 within 1e-5 of fp64), so the parity gates mean something on it."""
    import torch
    import torch.nn.functional as F
    np_sd = synth.vit_state_dict("vit_base", 16, seed=11, adversarial="outliers")
    sd = {k: torch.as_tensor(v).double() for k, v in np_sd.items()}
    z = np.load(os.path.join(golden_dir, "vit_base_out_tokens.npz"))
    x = vit_ref.preprocess_bgr_u8(z["frags"][:1]).double()
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    t = torch.cat((sd["cls_token"], t), dim=1) + sd["pos_embed"]
    for i in range(7):
        p = f"blocks.{i}."
        y = F.layer_norm(t, (768,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(1, 197, 3, 12, 64).permute(2, 0, 3, 1, 4)
        att = ((qkv[0] @ qkv[1].transpose(-2, -1)) / 8.0).softmax(dim=-1)
        if i == 0:
            assert float(att[0, 0].max(dim=-1).values.median()) > 0.9, "head 0 is not near-one-hot"
            assert float(att[0, 5].max(dim=-1).values.median()) < 0.1
        t = t + F.linear((att @ qkv[2]).transpose(1, 2).reshape(1, 197, 768), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y2 = F.layer_norm(t, (768,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        t = t + F.linear(F.gelu(F.linear(y2, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    chan = t[0].abs().max(dim=0).values
    top = torch.sort(chan, descending=True).values
    assert float(top[4] / chan.median()) > 50.0, "no outlier channels in the residual stream"
    ychan = y[0].abs().max(dim=0).values
    assert float(ychan.max() / ychan.median()) > 2.0 ** 9
    t32 = vit_ref.tokens(vit_ref.to_torch_state_dict(np_sd), z["frags"][:1], 12)
    t64 = vit_ref.forward_tokens(sd, x, 12).numpy()
    assert np.linalg.norm(t32 - t64) / np.linalg.norm(t64) < 1e-5
    quiet = np.abs(t64).max(axis=(0, 1)) < 10.0                       # ... also on the channels the outliers do not dominate
    assert quiet.sum() > 700 and np.linalg.norm((t32 - t64)[..., quiet]) / np.linalg.norm(t64[..., quiet]) < 1e-5
