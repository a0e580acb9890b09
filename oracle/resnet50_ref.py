"""Oracle (TEST INFRASTRUCTURE, not product): CPU fp32 restatement of the
ResNet-50 forward used by the reference extractors.

The reference calls ``torchvision.models.resnet50(pretrained=True)``
(/root/reference/src/extractor/visualise_resnet.py:21,
visualise_resnet_layer.py:20).  torchvision==0.17.2 (requirements.txt:119) is a
third-party dependency that is NOT vendored in the reference and NOT installed
in this image, so this file restates its published architecture (ResNet v1.5:
Bottleneck expansion 4, stride on the 3x3, BatchNorm eps 1e-5 in eval mode,
downsample = 1x1 conv + BN on block 0 of each stage) in plain
torch.nn.functional calls.  PARITY AGAINST THE REFERENCE: UNPINNED (no
torchvision, no weights); cross-checked against HuggingFace transformers'
independent ResNetModel in tests/test_oracle_resnet.py.

Call sites restated:
  visualise_resnet.py:40-50    preprocess (Resize identity at 224, ToTensor, Normalize)
  visualise_resnet.py:24-36    forward hook on one layer per forward
  main_fragment_layerstack.py:91-99  the 15 layer-stack taps / the avgpool tap
The state dict uses torchvision key names (conv1.weight, bn1.running_mean,
layer1.0.conv1.weight, layer1.0.downsample.0.weight, ...).
"""
import numpy as np
import torch
import torch.nn.functional as F

STAGES = [(1, 3, 64, 1), (2, 4, 128, 2), (3, 6, 256, 2), (4, 3, 512, 2)]  # (layer, blocks, width, stride)
TAPPED_BLOCKS = {1: (0, 1, 2), 2: (0, 1, 2, 3), 3: (0, 1, 2, 3), 4: (0, 1, 2)}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5


def preprocess_bgr_u8(frag_bgr_u8):
    """uint8 [N,224,224,3] BGR (as cv2 holds it) -> fp32 [N,3,224,224] RGB,
    /255 then (x-mean)/std.  The reference gets BGR->RGB for free from the PNG
    round trip (cv2.imwrite -> PIL.Image.open)."""
    x = torch.as_tensor(np.ascontiguousarray(frag_bgr_u8[..., ::-1]))
    x = x.permute(0, 3, 1, 2).to(torch.float32).div(255)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - mean) / std


def _bn(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def _bottleneck(x, sd, prefix, stride, has_down):
    out = F.relu(_bn(F.conv2d(x, sd[prefix + ".conv1.weight"]), sd, prefix + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[prefix + ".conv2.weight"], stride=stride, padding=1), sd, prefix + ".bn2"))
    out = _bn(F.conv2d(out, sd[prefix + ".conv3.weight"]), sd, prefix + ".bn3")
    if has_down:
        x = _bn(F.conv2d(x, sd[prefix + ".downsample.0.weight"], stride=stride), sd, prefix + ".downsample.1")
    return F.relu(out + x)


@torch.no_grad()
def forward_taps(sd, x):
    """x fp32 [N,3,224,224] -> (taps, avgpool): taps = ordered dict name->[N,C,H,W]
    for the 15 layer-stack taps (conv1 is the RAW conv output, before bn1),
    avgpool = [N,2048,1,1]."""
    taps = {}
    y = F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)
    taps["resnet50.conv1"] = y
    y = F.relu(_bn(y, sd, "bn1"))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    for layer, blocks, _width, stride in STAGES:
        for b in range(blocks):
            y = _bottleneck(y, sd, f"layer{layer}.{b}", stride if b == 0 else 1, b == 0)
            if b in TAPPED_BLOCKS[layer]:
                taps[f"resnet50.layer{layer}[{b}]"] = y
    avg = F.adaptive_avg_pool2d(y, 1)
    return taps, avg


def to_torch_state_dict(np_sd):
    return {k: torch.as_tensor(np.asarray(v)) for k, v in np_sd.items()}


def layer_stack_features(sd, frag_bgr_u8):
    """uint8 BGR fragments [N,224,224,3] -> fp32 [N,13120] (one forward per image,
    all taps; numerically what 15 hooked forwards give)."""
    taps, _ = forward_taps(sd, preprocess_bgr_u8(frag_bgr_u8))
    return torch.cat([t.mean(dim=(2, 3)) for t in taps.values()], dim=1).numpy()


def pool_features(sd, frag_bgr_u8):
    """uint8 BGR fragments [N,224,224,3] -> fp32 [N,2051]."""
    _, avg = forward_taps(sd, preprocess_bgr_u8(frag_bgr_u8))
    v = avg.flatten(1).numpy()
    stats = np.stack([v.mean(axis=1), v.max(axis=1), v.std(axis=1)], axis=1)
    return np.concatenate([v, stats], axis=1).astype(np.float32)
