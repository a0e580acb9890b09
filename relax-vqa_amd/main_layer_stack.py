"""Counterpart of the reference driver src/main_layer_stack.py (whole-frame features: every sampled frame is resized
to 224x224 and pushed through the ResNet-50 layer stack or the ViT), used by src/demo_test.py:81-87.

  get_deep_feature(network_name, video_name, image_path, qp)      four arguments (reference :81-112)
  process_video_feature(video_feature, network_name)              two arguments (reference :115-151)

The resize is the PIL-exact one on the GPU (`relax_resize_frames`): antialiased BILINEAR for ResNet-50, LANCZOS for the
ViT (src/extractor/visualise_resnet.py:40-47, visualise_vit_layer.py:466-469)."""
from . import main_fragment_layerstack as _ls


def get_deep_feature(network_name, video_name, image_path, qp):
    """image_path: an image file or a uint8 [H,W,3] BGR frame.  -> (png_path, npy_path, frame_npy)"""
    if network_name not in ("resnet50", "vit"):
        raise NotImplementedError(f"network {network_name!r} is out of scope (VGG-16 is an ablation backbone)")
    return _ls.get_deep_feature(network_name, video_name, image_path, qp, "layer_stack")


def process_video_feature(video_feature, network_name):
    """-> [T, 13120] (resnet50: per-tap spatial means) or [T, 2304] (vit: token mean | max | std)."""
    return _ls.process_video_feature(video_feature, network_name, "layer_stack")
