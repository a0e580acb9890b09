"""Dataset driver: BASELINE config 4 as written - a list of clips sharded over the ranks of one node, every rank pushing its
shard through the engine in batches, ONE all-gather of the per-clip vectors at the end.

The reference's counterpart is the per-video loop of its drivers (`for i in range(len(videodata))`,
src/main_fragment_layerstack.py:269-361: sample frames, fragments, features, `np.save` of the per-frame [T,F] array under
`video_{i+1}_{network}_feature_map_original.npy`) followed by src/data_processing/extract_npy2mat.py:117-130 (mean over the
frames of every file into one [n_videos, F] matrix, saved as a .mat keyed by the dataset name).  Here:

  rank r owns the contiguous block `shard_clips(n_clips, r, world)` of clip indices          (distributed.py)
  batches of `clips_per_step` clips go through RelaxEngine.clip_vectors / full_clip_vectors   (one pass of each backbone)
  optional: the per-frame [T,F] rows of every clip are written under the reference's file name (resume: skip_existing)
  the [n_local, F] means are all-gathered into the [n_clips, F] matrix every rank returns     (RCCL over xGMI; gloo on CPU)
  optional: rank 0 writes the .mat the reference's regression scripts read

Failure contract: a clip that cannot be loaded or extracted does not take its batch - or the run - down.  Its
row in the matrix is NaN and (clip index, message) goes into the returned error list (gathered over the ranks, sorted);
the reference's imputer zeroes NaN / inf before the regressor (src/model_regression.py:123-126), relax_mlp_head imputes
them with the training means.
"""
import os

import numpy as np
import torch

from . import distributed as rdist
from . import sampling
from .engine import LAYER_STACK_DIM, RN50_POOL_DIM

FULL_DIM = 35203


def feature_dim(engine, resnet=True, vit=True, full=False):
    if full:
        return FULL_DIM
    return (LAYER_STACK_DIM + RN50_POOL_DIM if resnet else 0) + (6 * engine.vit_dim if vit else 0)


def _check_clip(clip):
    """The argument errors the engine would raise for a whole batch, found per clip before the batch is assembled."""
    if not isinstance(clip, (torch.Tensor, np.ndarray)):
        raise TypeError(f"a clip must be a uint8 array [T,2,H,W,3], got {type(clip).__name__}")
    shape = tuple(clip.shape)
    if len(shape) != 5 or shape[1] != 2 or shape[4] != 3:
        raise ValueError(f"clip must be [T,2,H,W,3], got {shape}")
    if str(clip.dtype).replace("torch.", "") != "uint8":
        raise TypeError(f"clip must be uint8, got {clip.dtype}")
    if shape[0] == 0:
        raise ValueError("clip has no (frame, next) pair")
    if shape[2] < 16 or shape[3] < 16:
        raise ValueError(f"frames of {shape[3]}x{shape[2]} hold no 16x16 patch")


def extract_dataset_clips(clips, n_clips, engine, *, clips_per_step=16, resnet=True, vit=True, full=False, flow=True,
                          out_dir=None, network_name="resnet50", skip_existing=False, mat_path=None, data_name=None,
                          rank=None, world=None, group=None, timings=None):
    """clips: callable i -> uint8 [T,2,H,W,3] (device tensor, host tensor or ndarray; T and the resolution may differ from
    clip to clip), or a sequence indexed the same way.  Only this rank's shard is ever requested.
    -> (matrix fp32 [n_clips, F] on the engine's device, identical on every rank; errors [(clip index, message), ...]).

    out_dir: write each clip's per-frame rows [T, F] as `video_{i+1}_{network_name}_feature_map_original.npy`
             (sampling.feature_file_name); with skip_existing a clip whose file exists is not recomputed: its row is the mean
             of the stored rows (the resume the reference lacks).  Not available with full=True (whole-frame and fragment
             features have different frame counts there).
    mat_path / data_name: rank 0 saves the matrix as {data_name: float64 [n_clips, F]} (extract_npy2mat.py:79-84).
    timings: optional dict, receives 'extract_s' and 'all_gather_s' (device-synchronised wall times of the two phases)."""
    import time
    if rank is None or world is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
    if out_dir is not None and full:
        raise ValueError("per-frame files are written for the fragment features only (full=False)")
    get = clips if callable(clips) else clips.__getitem__
    mine = rdist.shard_clips(n_clips, rank, world)
    F = feature_dim(engine, resnet, vit, full)
    dev = engine.device
    local = torch.full((len(mine), F), float("nan"), dtype=torch.float32, device=dev)
    errors = []

    def run(batch):
        if full:
            return engine.full_clip_vectors(batch, flow=flow), None
        if out_dir is not None:
            return engine.clip_vectors(batch, resnet=resnet, vit=vit, per_frame=True)
        return engine.clip_vectors(batch, resnet=resnet, vit=vit), None

    def store(slots, idxs, vecs, frames):
        local[torch.as_tensor(slots, device=dev)] = vecs
        if frames is not None:
            for i, rows in zip(idxs, frames):
                sampling.save_clip_features(out_dir, i, network_name, rows.cpu().numpy())

    torch.cuda.synchronize(dev) if dev.type == "cuda" else None
    t0 = time.perf_counter()
    for lo in range(0, len(mine), max(int(clips_per_step), 1)):
        slots, idxs, batch = [], [], []
        for slot, i in enumerate(mine[lo:lo + clips_per_step], start=lo):
            try:
                if out_dir is not None and skip_existing:
                    path = os.path.join(out_dir, sampling.feature_file_name(i, network_name))
                    if os.path.exists(path):
                        rows = np.load(path)
                        if rows.ndim != 2 or rows.shape[1] != F:
                            raise ValueError(f"{path}: stored rows are {rows.shape}, expected [T,{F}]")
                        local[slot] = torch.from_numpy(rows.mean(axis=0).astype(np.float32)).to(dev)
                        continue
                clip = get(i)
                _check_clip(clip)
                slots.append(slot)
                idxs.append(i)
                batch.append(clip if isinstance(clip, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(clip)))
            except Exception as e:                      # noqa: BLE001 - the contract: the clip fails, the run goes on
                errors.append((i, f"{type(e).__name__}: {e}"))
        if not batch:
            continue
        try:
            vecs, frames = run(batch)
            store(slots, idxs, vecs, frames)
        except Exception as e0:                         # noqa: BLE001 - find the clip(s) that broke the batch, keep the others
            if len(batch) == 1:
                errors.append((idxs[0], f"{type(e0).__name__}: {e0}"))
                continue
            for slot, i, clip in zip(slots, idxs, batch):
                try:
                    vecs, frames = run([clip])
                    store([slot], [i], vecs, frames)
                except Exception as e:                  # noqa: BLE001
                    errors.append((i, f"{type(e).__name__}: {e}"))
    torch.cuda.synchronize(dev) if dev.type == "cuda" else None
    t1 = time.perf_counter()
    matrix = rdist.gather_clip_vectors(local, n_clips, rank, world, group)
    all_errors = sorted(e for part in rdist.gather_objects(errors, world, group) for e in part)
    torch.cuda.synchronize(dev) if dev.type == "cuda" else None
    t2 = time.perf_counter()
    if timings is not None:
        timings["extract_s"] = t1 - t0
        timings["all_gather_s"] = t2 - t1
    if mat_path is not None and rank == 0:
        sampling.save_mat(mat_path, data_name or "features", matrix.cpu().numpy().astype(np.float64))
    return matrix, all_errors
