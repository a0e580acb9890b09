"""Counterpart of src/extractor/visualise_resnet_layer.py (reference): one hooked ResNet-50 layer, used with
'resnet50.avgpool' for the 'pool' features (src/main_fragment_layerstack.py:97-99)."""
import os

import numpy as np
import torch

from .. import runtime
from .visualise_resnet import LAYER_INDEX, _frame_number


class PoolActivation(np.ndarray):
    """ndarray [2048,1,1] like the reference returns, plus `.pooled`: the fp32 [2051] vector (avgpool + mean/max/std)."""
    pooled = None


def process_fragment_array(frag_bgr_u8, layer_name="resnet50.avgpool"):
    eng = runtime.ensure_resnet50()
    x = torch.from_numpy(np.ascontiguousarray(frag_bgr_u8))
    if layer_name == "resnet50.avgpool":
        _, pool = eng.resnet50_features(x, layer_stack=False, pool=True)
        p = pool[0].cpu().numpy()
        out = p[:2048].reshape(2048, 1, 1).view(PoolActivation)
        out.pooled = p
        return out
    if layer_name in LAYER_INDEX:
        i = LAYER_INDEX[layer_name]
        _, _, taps = eng.resnet50_features(x, layer_stack=False, pool=False, taps=[i])
        return taps[i][0].cpu().numpy()
    raise ValueError(f"unknown ResNet-50 layer selector {layer_name!r}")


def process_video_frame(video_name, image_path, layer_name, qp):
    filename = os.path.basename(image_path)
    frame_number = _frame_number(filename)
    img = runtime.to_model_input(runtime.read_image_bgr(image_path), "resnet50")
    arr = process_fragment_array(img, layer_name)
    combined = "resnet50_feature_map_original" if qp == "original_ugc" else f"resnet50_feature_map_qp_{qp}"
    return arr, f"../features/resnet50/{video_name}/frame_{frame_number}_{combined}.npy"
