"""The ResNet-50 restatement has no reference-side pin (torchvision absent); cross-check it against
HuggingFace transformers' independent implementation of the same architecture."""
import numpy as np
import pytest
import torch

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import synth
from oracle import fragment_ref, pooling_ref, resnet50_ref


def _hf_model(sd):
    tr = pytest.importorskip("transformers")
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048],
                          depths=[3, 4, 6, 3], layer_type="bottleneck", hidden_act="relu",
                          downsample_in_first_stage=False, downsample_in_bottleneck=False)
    m = tr.ResNetModel(cfg).eval()
    hf = {}

    def put(dst, src_conv, src_bn):
        hf[dst + ".convolution.weight"] = sd[src_conv + ".weight"]
        for a in ("weight", "bias", "running_mean", "running_var"):
            hf[dst + ".normalization." + a] = sd[src_bn + "." + a]

    put("embedder.embedder", "conv1", "bn1")
    for s, (layer, blocks, _w, _st) in enumerate(resnet50_ref.STAGES):
        for b in range(blocks):
            p, q = f"layer{layer}.{b}", f"encoder.stages.{s}.layers.{b}"
            for j in range(3):
                put(f"{q}.layer.{j}", f"{p}.conv{j + 1}", f"{p}.bn{j + 1}")
            if b == 0:
                put(f"{q}.shortcut", f"{p}.downsample.0", f"{p}.downsample.1")
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    return m


def test_restatement_matches_independent_implementation():
    sd = resnet50_ref.to_torch_state_dict(synth.resnet50_state_dict(seed=7))
    o, n = synth.synthetic_pair(240, 320, 77)
    frag = fragment_ref.fragment_pair(o, n)["ori_frag"][None]
    x = resnet50_ref.preprocess_bgr_u8(frag)
    taps, avg = resnet50_ref.forward_taps(sd, x)
    m = _hf_model(sd)
    with torch.no_grad():
        out = m(x, output_hidden_states=True)
    # hidden_states: embedder output (post maxpool), then each stage output
    stage_last = ["resnet50.layer1[2]", "resnet50.layer2[3]", None, "resnet50.layer4[2]"]
    for s, name in enumerate(stage_last):
        if name is None:
            continue  # layer3's last block (index 5) is not a tap
        np.testing.assert_allclose(taps[name].numpy(), out.hidden_states[s + 1].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(avg.numpy(), out.pooler_output.numpy(), rtol=1e-4, atol=1e-4)


def test_feature_shapes_and_tap_order():
    sd = resnet50_ref.to_torch_state_dict(synth.resnet50_state_dict(seed=7))
    frag = np.random.default_rng(0).integers(0, 256, (1, 224, 224, 3), dtype=np.uint8)
    taps, avg = resnet50_ref.forward_taps(sd, resnet50_ref.preprocess_bgr_u8(frag))
    assert list(taps.keys()) == pooling_ref.RESNET50_TAPS
    assert [t.shape[1] for t in taps.values()] == pooling_ref.RESNET50_TAP_CHANNELS
    assert taps["resnet50.conv1"].shape[2:] == (112, 112) and avg.shape == (1, 2048, 1, 1)
    ls = resnet50_ref.layer_stack_features(sd, frag)
    pool = resnet50_ref.pool_features(sd, frag)
    assert ls.shape == (1, 13120) and pool.shape == (1, 2051)
    # the pool vector's first 2048 entries are the last layer-stack tap
    np.testing.assert_allclose(pool[0, :2048], ls[0, -2048:], rtol=1e-5, atol=1e-6)
    # reference pooling on the hooked arrays gives the same numbers
    ref_ls = pooling_ref.layer_stack_vector({k: v[0].numpy() for k, v in taps.items()})
    np.testing.assert_allclose(ls[0], ref_ls, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pool[0], pooling_ref.resnet_pool_vector(avg[0].numpy()), rtol=1e-5, atol=1e-6)
