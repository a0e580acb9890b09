"""Oracle (TEST INFRASTRUCTURE, not product): the per-clip CPU pipeline, in the two schedules of
BASELINE.md §4.  Used as the checker for whole-clip parity and as the timed ``cpu_baseline`` ("port")
of bench.py.  In-memory frames in, feature rows out; decode / PNG / ffmpeg excluded on both sides.

  faithful : what the reference executes per (frame, next) pair
             - python double-loop patch scoring               (src/main_fragment_layerstack.py:177-189)
             - 15 separate hooked ResNet-50 forwards for the layer-stack fragment + 1 avgpool forward
               for the residual fragment                       (src/extractor/visualise_resnet.py:83-98)
             - ViT rebuilt for every call                      (src/main_fragment_layerstack.py:118;
               the hub weight load is replaced by an in-memory state-dict copy)
  dedup    : one forward per image with all taps, vectorised scoring, models built once.
Batch size is 1 in both, as in the reference.
"""
import numpy as np
import torch

from . import fragment_ref, pooling_ref, resnet50_ref, vit_ref


def _rn_layer_stack_faithful(sd, frag):
    x = resnet50_ref.preprocess_bgr_u8(frag[None])
    feats = []
    for name in pooling_ref.RESNET50_TAPS:           # one full forward per hooked layer
        taps, _ = resnet50_ref.forward_taps(sd, x)
        feats.append(taps[name][0].numpy().mean(axis=(1, 2)))
    return np.hstack(feats)


def _vit_pool_faithful(vit_np_sd, frag, heads):
    sd = {k: torch.tensor(v) for k, v in vit_np_sd.items()}   # fresh copy per call = model rebuild + load
    return vit_ref.pool_features(sd, frag[None], heads)[0]


def clip_features(frames, rn_np_sd, vit_np_sd=None, heads=12, schedule="dedup"):
    """frames uint8 [T,2,H,W,3] -> dict(resnet [T,15171], vit [T,4608] (if vit weights), positions)"""
    assert schedule in ("dedup", "faithful")
    rn_sd = resnet50_ref.to_torch_state_dict(rn_np_sd)
    vit_sd = vit_ref.to_torch_state_dict(vit_np_sd) if vit_np_sd is not None else None
    rows_rn, rows_vit, positions = [], [], []
    for t in range(frames.shape[0]):
        fp = fragment_ref.fragment_pair(frames[t, 0], frames[t, 1], loop_score=(schedule == "faithful"))
        positions.append(fp["positions"])
        ori, res = fp["ori_frag"], fp["diff_frag"]
        if schedule == "faithful":
            ls = _rn_layer_stack_faithful(rn_sd, ori)
            pool = resnet50_ref.pool_features(rn_sd, res[None])[0]
        else:
            ls = resnet50_ref.layer_stack_features(rn_sd, ori[None])[0]
            pool = resnet50_ref.pool_features(rn_sd, res[None])[0]
        rows_rn.append(np.concatenate([ls, pool]))
        if vit_np_sd is not None:
            if schedule == "faithful":
                v = np.concatenate([_vit_pool_faithful(vit_np_sd, ori, heads), _vit_pool_faithful(vit_np_sd, res, heads)])
            else:
                v = np.concatenate([vit_ref.pool_features(vit_sd, ori[None], heads)[0],
                                    vit_ref.pool_features(vit_sd, res[None], heads)[0]])
            rows_vit.append(v)
    out = dict(resnet=np.stack(rows_rn).astype(np.float32), positions=positions)
    if rows_vit:
        out["vit"] = np.stack(rows_vit).astype(np.float32)
    return out
