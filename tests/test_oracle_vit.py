import os

import numpy as np
import pytest

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import synth
from oracle import vit_ref


@pytest.mark.parametrize("name,heads", [("vit_tiny", 3), ("vit_base", 12)])
def test_vit_tokens_match_reference_golden(golden_dir, name, heads):
    z = np.load(os.path.join(golden_dir, f"{name}_tokens.npz"))
    sd_np = synth.vit_state_dict(name, 16, seed=11)
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
