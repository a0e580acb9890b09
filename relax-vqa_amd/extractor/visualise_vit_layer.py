"""Counterpart of src/extractor/visualise_vit_layer.py (reference): DINO ViT patch tokens.

Reference: VitGenerator(name_model, patch_size, device, evaluate=True, random=False, verbose=False) builds the model
and loads hub weights (:263-329); process_video_frame(image_path, video_name, qp, model, patch_size, device) returns
the final-norm patch tokens [196, dim] (:447-500).  The reference rebuilds the generator for every frame
(src/main_fragment_layerstack.py:118); here the weights live in the engine and the generator is a light handle."""
import os

import numpy as np
import torch

from .. import runtime, synth
from .visualise_resnet import _frame_number


class VitGenerator(object):
    def __init__(self, name_model, patch_size, device=None, evaluate=True, random=False, verbose=False):
        if name_model not in ("vit_tiny", "vit_small", "vit_base"):
            raise ValueError(f"No model found with {name_model}")   # the reference raises a bare string here (:291)
        if patch_size != 16:
            raise NotImplementedError("only patch_size 16 (197 tokens at 224x224) is built")
        self.name_model = name_model
        self.patch_size = patch_size
        self.device = device
        self.evaluate = evaluate
        self.verbose = verbose
        if random:
            runtime.set_weights(vit=synth.vit_state_dict(name_model), vit_name=name_model)
        else:
            runtime.ensure_vit(name_model)

    def tokens(self, frag_bgr_u8):
        eng = runtime.ensure_vit(self.name_model)
        t, _ = eng.vit_features(torch.from_numpy(np.ascontiguousarray(frag_bgr_u8)), tokens=True, pooled=False)
        return t

    def __call__(self, frag_bgr_u8):
        """-> (None, tokens [N,196,dim]); the cls token is not part of the hot path."""
        return None, self.tokens(frag_bgr_u8)


def process_fragment_array(frag_bgr_u8, model):
    return model.tokens(frag_bgr_u8)[0].cpu().numpy()


def process_video_frame(image_path, video_name, qp, model, patch_size, device):
    filename = os.path.basename(image_path)
    frame_number = _frame_number(filename)
    img = runtime.to_model_input(runtime.read_image_bgr(image_path), "vit")
    feats = process_fragment_array(img, model)
    combined = "vit_feature_map_original" if qp == "original" else f"vit_feature_map_qp_{qp}"
    return feats, f"../features/vit/{video_name}/frame_attention_{frame_number}_{combined}.npy"
