"""Counterpart of the reference driver src/main_residual_fragment.py (the residual-fragment-only variant): same
function names, argument order and return arity as that file, running on the HIP engine.

Differences from main_fragment_layerstack in the reference, kept here:
  * extract_important_patches returns only the fragment (reference :187-204), process_patches only the path (:206-214);
  * process_video_feature takes two arguments and always pools the `pool` way (:118-156);
  * get_deep_feature knows layer_name 'pool' and 'last_layer' (:83-115).
The reference passes the fragment to the next stage through a PNG on disk (`cv2.imwrite`, :213); here the fragment is
kept in memory under the same path string, so `get_deep_feature(..., residual_frag_path, ...)` finds it without I/O.
"""
import numpy as np

from . import main_fragment_layerstack as _ls
from . import runtime
from .extractor import visualise_resnet_layer, visualise_vit_layer
from .main_fragment_layerstack import flow_to_rgb, get_patch_diff  # noqa: F401  (same behaviour in both drivers)

_fragments_by_path = {}     # path string -> uint8 [224,224,3]; bounded: the reference consumes each one right away
_MAX_HELD = 256


def extract_important_patches(residual_frame, diff=None, patch_size=16, target_size=224, top_n=196):
    """-> fragment uint8 [224,224,3] only (reference :187-204)."""
    return _ls.extract_important_patches(residual_frame, diff, patch_size, target_size, top_n)[0]


def process_patches(original_path, residual_name, residual, patch_size=16, target_size=224, top_n=196):
    """-> residual_frag_path (reference :206-214).  The fragment is held in memory under that name instead of a PNG."""
    path, frag, _ = _ls.process_patches(original_path, residual_name, residual, patch_size, target_size, top_n)
    if len(_fragments_by_path) >= _MAX_HELD:
        _fragments_by_path.pop(next(iter(_fragments_by_path)))
    _fragments_by_path[path] = frag
    return path


def get_deep_feature(network_name, video_name, image_path, qp, layer_name):
    """-> (png_path, npy_path, frame_npy) (reference :83-115); image_path may be a path returned by process_patches,
    a real image file, or a uint8 [224,224,3] array."""
    png_path = f"../visualisation/{network_name}/{video_name}/"
    npy_path = f"../features/{network_name}/{video_name}/"
    if isinstance(image_path, str):
        image = _fragments_by_path.pop(image_path, None)
        if image is None:
            image = runtime.read_image_bgr(image_path)
    else:
        image = image_path
    image = runtime.to_model_input(image, network_name)
    if network_name == "resnet50":
        if layer_name == "pool":
            frame_npy = visualise_resnet_layer.process_fragment_array(image, "resnet50.avgpool")
        elif layer_name == "last_layer":
            frame_npy = visualise_resnet_layer.process_fragment_array(image, "resnet50.layer4[2]")
        else:
            raise ValueError(f"unknown layer_name {layer_name!r}")      # the reference hits an unbound local here
    elif network_name == "vit":
        model = visualise_vit_layer.VitGenerator("vit_base", 16, None, evaluate=True, random=False, verbose=False)
        frame_npy = visualise_vit_layer.process_fragment_array(image, model)
    else:
        raise NotImplementedError(f"network {network_name!r} is out of scope (VGG-16 is an ablation backbone)")
    return png_path, npy_path, frame_npy


def process_video_feature(video_feature, network_name):
    """list of per-frame activations -> [T, 2304] (vit) or [T, 2051] (resnet50 pool) (reference :118-156)."""
    if network_name != "vit":
        for frame in video_feature:
            if getattr(frame, "pooled", None) is None and np.squeeze(frame).ndim != 1:
                # 'last_layer' activations [2048,7,7] go through np.squeeze + axis-0 statistics in the reference and
                # produce a ragged hstack; nothing downstream uses that, so it is not reproduced
                raise NotImplementedError("process_video_feature pools 'pool' activations; 'last_layer' is a "
                                          "visualisation tap")
    return _ls.process_video_feature(video_feature, network_name, "pool")
