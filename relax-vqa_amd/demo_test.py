"""Counterpart of the reference's end-to-end demo (src/demo_test.py:51-219) on in-memory frames: the full 35203-d
clip vector (whole-frame + fragment features of both backbones) through imputer, scaler and the MLP head, all on the
GPU.  ffmpeg sampling (src/extractor/vf_extract.py) is out of scope: the caller supplies the sampled (frame, next)
pairs, uint8 [T,2,H,W,3] BGR.  With flow=True (default) the residual fragment is the reference's 50/50 merge of the
frame-difference and optical-flow fragments (Farneback + flow_to_rgb on the GPU)."""
import numpy as np
import torch

from . import runtime


def load_head(state_dict, imputer, scaler):
    """state_dict: the reference Mlp's weights (after torch.load); imputer / scaler: the sklearn objects of
    model/scaler/{video_type}_imputer.pkl / _scaler.pkl, or (statistics, scale, min) arrays."""
    if hasattr(scaler, "scale_"):
        scale, mn = scaler.scale_, scaler.min_
    else:
        scale, mn = scaler
    stats = getattr(imputer, "statistics_", imputer)
    sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()}
    runtime.get_engine().load_mlp_head(sd, scale, mn, stats)


def evaluate_video_quality(frames, video_type="konvid_1k", is_finetune=False, flow_images=None, flow=True):
    """frames uint8 [T,2,H,W,3] (numpy or device tensor) -> predicted quality score (float).
    Rescaling rule of src/demo_test.py:211-219: non-fine-tuned models on youtube_ugc / konvid_1k map 0-100 to 1-5."""
    eng = runtime.ensure_vit("vit_base")
    runtime.ensure_resnet50()
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(frames)
    frames = frames.to(eng.device)
    flow_img = None if flow_images is None else torch.as_tensor(flow_images).to(eng.device)
    vec = eng.full_clip_vector(frames, flow_images=flow_img, flow=flow)
    pred = float(eng.mlp_head(vec[None])[0].item())
    if not is_finetune and video_type in ("youtube_ugc", "konvid_1k"):
        pred = (pred / 100) * 4 + 1
    return pred
