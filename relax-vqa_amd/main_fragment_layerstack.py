"""Counterpart of the reference drivers src/main_fragment_layerstack.py / src/main_residual_fragment.py /
src/main_fragment_pool.py: same function names and argument meaning, arrays instead of PNG files where the
reference only used files as glue, every function running on the HIP engine (no CPU fallback).

  get_patch_diff, extract_important_patches, get_original_frame_patches, process_patches, merge_fragments,
  concatenate_features, get_deep_feature, process_video_feature        (reference :83-248)
  calc_optical_flow_farneback, flow_to_rgb                              (reference :313-316, :162-175)
"""
import numpy as np
import torch

from . import runtime
from .extractor import visualise_resnet, visualise_resnet_layer, visualise_vit_layer

ALL_LAYERS = list(visualise_resnet.LAYER_INDEX)


def _u8(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected uint8 [H,W,3], got {a.dtype} {a.shape}")
    return a


def _check_geometry(patch_size, target_size, top_n):
    if patch_size != 16 or target_size != 224 or top_n > 196:
        raise NotImplementedError("the HIP fragment stage is built for patch_size=16, target_size=224, top_n<=196")


def get_patch_diff(residual_frame, patch_size=16):
    """-> float64 [H//16, W//16] patch sums (reference :177-189)."""
    _check_geometry(patch_size, 224, 196)
    out = runtime.get_engine().fragment_image(torch.from_numpy(_u8(residual_frame)[None]), want_scores=True)
    return out["scores"][0].cpu().numpy().astype(np.float64)


def extract_important_patches(residual_frame, diff=None, patch_size=16, target_size=224, top_n=196):
    """-> (fragment uint8 [224,224,3], positions list of (y,x)) (reference :191-210).  `diff` is recomputed on the
    GPU (score, selection and gather are one fused pass); it is accepted for signature compatibility."""
    _check_geometry(patch_size, target_size, top_n)
    out = runtime.get_engine().fragment_image(torch.from_numpy(_u8(residual_frame)[None]), top_n=top_n)
    n = int(out["counts"][0])
    pos = out["positions"][0, :n].cpu().numpy()
    return out["frag"][0].cpu().numpy(), [(int(y), int(x)) for y, x in pos]


def get_original_frame_patches(original_frame, positions, patch_size=16, target_size=224):
    _check_geometry(patch_size, target_size, len(positions))
    pos = torch.full((1, 196, 2), -1, dtype=torch.int32)
    if len(positions):
        pos[0, :len(positions)] = torch.as_tensor(np.asarray(positions, dtype=np.int32))
    cnt = torch.tensor([len(positions)], dtype=torch.int32)
    return runtime.get_engine().gather_patches(torch.from_numpy(_u8(original_frame)[None]), pos, cnt)[0].cpu().numpy()


def process_patches(original_path, residual_name, residual, patch_size=16, target_size=224, top_n=196):
    """-> (fragment path, fragment, positions) (reference :232-240; nothing is written to disk)."""
    frag, positions = extract_important_patches(residual, None, patch_size, target_size, top_n)
    suffix = "_residual_imp.png" if residual_name == "frame_diff" else "_residual_of_imp.png"
    return original_path.replace(".png", suffix), frag, positions


def fragment_pair(img_original, img_next, top_n=196):
    """Fused form of cv2.absdiff + process_patches('frame_diff') + get_original_frame_patches (reference :302-310).
    -> (diff_fragment, original_fragment, positions)"""
    frames = torch.from_numpy(np.stack([_u8(img_original), _u8(img_next)])[None])
    out = runtime.get_engine().fragment_pairs(frames, top_n=top_n)
    n = int(out["counts"][0])
    pos = [(int(y), int(x)) for y, x in out["positions"][0, :n].cpu().numpy()]
    return out["diff_frag"][0].cpu().numpy(), out["ori_frag"][0].cpu().numpy(), pos


def merge_fragments(diff_fragment, flow_fragment):
    return runtime.get_engine().merge_fragments(torch.from_numpy(_u8(diff_fragment)),
                                                torch.from_numpy(_u8(flow_fragment))).cpu().numpy()


def calc_optical_flow_farneback(img_original, img_next):
    """cv2.calcOpticalFlowFarneback(gray(orig), gray(next), None, 0.5, 3, 15, 3, 5, 1.2, 0) on the GPU (reference
    :313-315; BGR->gray included).  -> float32 [H,W,2]"""
    frames = torch.from_numpy(np.stack([_u8(img_original), _u8(img_next)])[None])
    flow, _ = runtime.get_engine().optical_flow(frames, want_flow=True, want_image=False)
    return flow[0].cpu().numpy()


def flow_to_rgb(flow):
    """float32 [H,W,2] -> uint8 [H,W,3] flow visualisation (reference :162-175; BGR despite the name)."""
    f = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32))[None]
    return runtime.get_engine().flow_to_rgb(f)[0].cpu().numpy()


def concatenate_features(original_feature, residual_feature):
    return np.concatenate((original_feature, residual_feature), axis=-1)


def get_deep_feature(network_name, video_name, image, qp, layer_name):
    """image: a PNG path (reference signature, :83-121) or a uint8 [224,224,3] BGR fragment array.
    -> (png_path, npy_path, frame_npy) with frame_npy as the reference returns it (dict of taps / [2048,1,1] /
    [196,768]) carrying the GPU-pooled vector as `.pooled` where one exists."""
    png_path = f"../visualisation/{network_name}/{video_name}/"
    npy_path = f"../features/{network_name}/{video_name}/"
    if isinstance(image, str):
        image = runtime.read_image_bgr(image)
    image = runtime.to_model_input(image, network_name)   # whole frames: PIL-exact resize on the GPU (main_layer_stack.py)
    if network_name == "resnet50":
        if layer_name == "layer_stack":
            frame_npy = visualise_resnet.process_fragment_array(image, ALL_LAYERS)
        elif layer_name == "pool":
            frame_npy = visualise_resnet_layer.process_fragment_array(image, "resnet50.avgpool")
        else:
            raise ValueError(f"unknown layer_name {layer_name!r}")
    elif network_name == "vit":
        model = visualise_vit_layer.VitGenerator("vit_base", 16, None, evaluate=True, random=False, verbose=False)
        frame_npy = visualise_vit_layer.process_fragment_array(image, model)
    else:
        raise NotImplementedError(f"network {network_name!r} is out of scope (VGG-16 is an ablation backbone)")
    return png_path, npy_path, frame_npy


def _gap_on_gpu(arr_chw):
    eng = runtime.get_engine()
    c = arr_chw.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(arr_chw.reshape(c, -1).T))[None].cuda()   # [1, HW, C]
    if c % 64:
        raise ValueError("layer_stack pooling expects channel counts that are multiples of 64")
    return eng.op_gap(x)[0].cpu().numpy()


def process_video_feature(video_feature, network_name, layer_name="pool"):
    """list of per-frame activations -> ndarray [T, F] (reference :124-160; 2-argument form of
    main_residual_fragment.py:118).  Uses the vectors pooled on the GPU when the activations came from
    get_deep_feature; plain arrays are reduced on the GPU here."""
    rows = []
    for frame in video_feature:
        pooled = getattr(frame, "pooled", None)
        if network_name == "vit":
            eng = runtime.get_engine()
            t = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.float32)).cuda()
            rows.append(_vit_stats(eng, t))
        elif layer_name == "layer_stack":
            rows.append(pooled if pooled is not None else np.hstack([_gap_on_gpu(a) for a in frame.values()]))
        else:
            if pooled is None:
                raise NotImplementedError("pool features need the activation returned by get_deep_feature(..., 'pool')")
            rows.append(pooled)
    return np.array(rows)


def _vit_stats(eng, tokens):
    """tokens [196, dim] (final-norm patch tokens) -> [3*dim] mean | max | population std, on the engine."""
    return eng.op_token_stats(tokens[None].contiguous())[0].cpu().numpy()
