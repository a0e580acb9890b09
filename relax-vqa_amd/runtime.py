"""Process-wide engine + weights, the counterpart of the reference's import-time globals
(`resnet50 = models.resnet50(pretrained=True).to(device)`, src/extractor/visualise_resnet.py:17-21).
Created lazily, never at import.  Weights: a torchvision / DINO state-dict file named by
RELAX_RESNET50_WEIGHTS / RELAX_VIT_WEIGHTS.  The reference always runs pretrained weights, so a missing
variable is an error; the deterministic synthetic weights (there is no network in the build environment)
are used only when RELAX_ALLOW_SYNTHETIC_WEIGHTS=1 says so, or when injected explicitly with set_weights()."""
import logging
import os

import numpy as np

from . import synth

log = logging.getLogger("relax_vqa_amd")
_state = {"engine": None, "rn": False, "vit": None}


_WRAPPER_KEYS = ("state_dict", "model", "teacher", "student")   # torch.save({"state_dict": ...}), DINO full checkpoints
_KEY_PREFIXES = ("module.", "backbone.")                        # DataParallel / DINO's MultiCropWrapper


def _load_file(path):
    """A checkpoint file -> {torchvision / DINO key: fp32 array}.  The reference loads `models.resnet50(pretrained=True)`
    (src/extractor/visualise_resnet.py:21) and the DINO hub file with `load_state_dict(strict=True)`
    (src/extractor/visualise_vit_layer.py:326-328): plain state dicts.  Files saved around them are accepted too: one
    level of {"state_dict" | "model" | "teacher" | "student": ...} wrapping and `module.` / `backbone.` key prefixes."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise RuntimeError(f"{path}: expected a state dict, found {type(sd).__name__}")
    for k in _WRAPPER_KEYS:
        if isinstance(sd.get(k), dict):
            sd = sd[k]
            break
    out = {}
    for k, v in sd.items():
        if not hasattr(v, "shape"):
            continue                                     # epoch counters and the like
        stripped = True
        while stripped:
            stripped = False
            for p in _KEY_PREFIXES:
                if k.startswith(p):
                    k, stripped = k[len(p):], True
        out[k] = v.numpy() if hasattr(v, "numpy") else np.asarray(v)
    return out


def get_engine(device=None):
    """The process-wide engine.  One process per GPU: the device defaults to LOCAL_RANK (torchrun)."""
    if _state["engine"] is None:
        from .engine import RelaxEngine
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        _state["engine"] = RelaxEngine(device)
    return _state["engine"]


def _weights_or_synthetic(env_var, what, make_synthetic):
    path = os.environ.get(env_var)
    if path:
        return _load_file(path)
    if os.environ.get("RELAX_ALLOW_SYNTHETIC_WEIGHTS") == "1":
        log.warning("%s not set: using synthetic %s weights (RELAX_ALLOW_SYNTHETIC_WEIGHTS=1)", env_var, what)
        return make_synthetic()
    raise RuntimeError(
        f"{env_var} is not set: point it at the pretrained {what} state dict (the reference loads pretrained weights), "
        "inject weights with runtime.set_weights(), or set RELAX_ALLOW_SYNTHETIC_WEIGHTS=1 to run on random-init weights")


def ensure_resnet50():
    eng = get_engine()
    if not _state["rn"]:
        sd = _weights_or_synthetic("RELAX_RESNET50_WEIGHTS", "ResNet-50", synth.resnet50_state_dict)
        eng.load_resnet50(sd)
        _state["rn"] = True
    return eng


def ensure_vit(name_model="vit_base"):
    eng = get_engine()
    if _state["vit"] != name_model:
        sd = _weights_or_synthetic("RELAX_VIT_WEIGHTS", name_model, lambda: synth.vit_state_dict(name_model))
        eng.load_vit(sd, name_model)
        _state["vit"] = name_model
    return eng


def set_weights(resnet50=None, vit=None, vit_name="vit_base"):
    """Explicit weight injection (tests, real checkpoints already in memory)."""
    eng = get_engine()
    if resnet50 is not None:
        eng.load_resnet50(resnet50)
        _state["rn"] = True
    if vit is not None:
        eng.load_vit(vit, vit_name)
        _state["vit"] = vit_name
    return eng


def reset_weights():
    """Forget which weights are loaded: the next ensure_resnet50() / ensure_vit() reads RELAX_*_WEIGHTS again."""
    _state["rn"] = False
    _state["vit"] = None


def read_image_bgr(image_path):
    """PNG/JPEG -> uint8 [H,W,3] BGR (what cv2.imread returns, src/main_fragment_layerstack.py:295)."""
    from PIL import Image
    rgb = np.asarray(Image.open(image_path).convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


def to_model_input(img_bgr, network_name):
    """uint8 [H,W,3] -> uint8 [224,224,3] the way the reference's extractors do it: a fragment passes through; a whole
    frame is resized on the GPU with PIL-exact BILINEAR (ResNet-50, torchvision Resize on a PIL image,
    src/extractor/visualise_resnet.py:40-47) or LANCZOS (ViT, src/extractor/visualise_vit_layer.py:466-469)."""
    if img_bgr.shape[:2] == (224, 224):
        return img_bgr
    eng = get_engine()
    vit = network_name == "vit"
    bil, lan = eng.resize_frames(img_bgr[None], bilinear=not vit, lanczos=vit)
    return (lan if vit else bil)[0].cpu().numpy()
