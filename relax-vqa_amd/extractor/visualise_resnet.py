"""Counterpart of src/extractor/visualise_resnet.py (reference): layer-stack activations of ResNet-50.

Reference: process_video_frame(video_name, image_path, all_layers, qp) loops over the 15 layer names and runs one
full forward per layer through a forward hook (:62-109, :24-36).  Here ONE forward on the GPU yields all taps."""
import os

import numpy as np
import torch

from .. import runtime

# selector vocabulary of the reference (src/main_fragment_layerstack.py:91-95); the reference resolves these with
# eval() (visualise_resnet.py:54) - a lookup table here
LAYER_INDEX = {name: i for i, name in enumerate([
    "resnet50.conv1",
    "resnet50.layer1[0]", "resnet50.layer1[1]", "resnet50.layer1[2]",
    "resnet50.layer2[0]", "resnet50.layer2[1]", "resnet50.layer2[2]", "resnet50.layer2[3]",
    "resnet50.layer3[0]", "resnet50.layer3[1]", "resnet50.layer3[2]", "resnet50.layer3[3]",
    "resnet50.layer4[0]", "resnet50.layer4[1]", "resnet50.layer4[2]"])}


class LayerStackActivations(dict):
    """dict layer_name -> ndarray [C,H,W] like the reference returns, plus `.pooled`: the fp32 [13120] spatial means
    already computed on the GPU (process_video_feature uses it instead of re-reducing 23.7 MB on the host)."""
    pooled = None


def _frame_number(filename):
    # same naming rules as the reference (:63-79): fragment files carry their suffixes into the frame id
    # the reference splits the whole file name at '_' and strips the extension from the LAST part only (:64-79), so a
    # dot earlier in the base name ("clip.v2_12_next.png") survives
    parts = filename.split("_")
    parts[-1] = parts[-1].split(".")[0]
    for marker, n in (("residual_of_imp", 4), ("residual_merged_frag", 4), ("residual_of", 3), ("residual_imp", 3),
                      ("ori_frag", 3)):
        if marker in filename:
            return "_".join(parts[-n:])
    if "residual" in filename or "next" in filename or "ori" in filename:
        return "_".join(parts[-2:])
    return int(parts[-1])


def process_fragment_array(frag_bgr_u8, all_layers):
    """Array form: uint8 [224,224,3] BGR -> LayerStackActivations."""
    eng = runtime.ensure_resnet50()
    idx = [LAYER_INDEX[name] for name in all_layers]
    ls, _, taps = eng.resnet50_features(torch.from_numpy(np.ascontiguousarray(frag_bgr_u8)), layer_stack=True,
                                        pool=False, taps=idx)
    out = LayerStackActivations()
    for name, i in zip(all_layers, idx):
        out[name] = taps[i][0].cpu().numpy()
    out.pooled = ls[0].cpu().numpy()
    return out


def process_video_frame(video_name, image_path, all_layers, qp):
    filename = os.path.basename(image_path)
    frame_number = _frame_number(filename)
    img = runtime.to_model_input(runtime.read_image_bgr(image_path), "resnet50")
    activations = process_fragment_array(img, all_layers)
    combined = "resnet50_feature_map_original" if qp == "original_ugc" else f"resnet50_feature_map_qp_{qp}"
    return activations, f"../features/resnet50/{video_name}/frame_{frame_number}_{combined}.npy"
