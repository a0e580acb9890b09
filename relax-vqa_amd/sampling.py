"""Frame-sampling semantics and on-disk feature formats of the reference, without ffmpeg: the caller
decodes the video (or already holds the frames) and these helpers reproduce which frames the reference's ffmpeg
`select` filters pick, how they are paired, and how per-clip features are named and packed.

  frame_interval          src/main_fragment_layerstack.py:274-277   (2 samples per second)
  sampled_frame_indices   src/video_frames_extract.py:12-20 (not(mod(n,k))), :61-66 (not(mod(n-1,k))), pairing by sorted
                          index with zip truncation src/main_fragment_layerstack.py:283-293
  frame_pair_paths / load_clip_from_frames   src/main_fragment_layerstack.py:283-296: the sampled frames the reference's ffmpeg step left on disk
                          (`{video}_{n}.png`, `{video}_{n}_next.png`), paired by sorted index, read as cv2.imread reads them (uint8 BGR)
  feature_file_name       src/main_fragment_layerstack.py:67-68,349-354  video_{i+1}_{network}_feature_map_original.npy
  features_matrix / save_mat  src/data_processing/extract_npy2mat.py:117-130, 79-84 (np.mean over frames; .mat key = dataset)
"""
import math
import os
import threading

import numpy as np


def frame_interval(framerate):
    return math.ceil(framerate / 2) if framerate < 2 else int(framerate / 2)


def sampled_frame_indices(n_frames, interval):
    """-> (sampled, following, pairs): frames n with n % k == 0, frames n with (n-1) % k == 0, and the (frame, next)
    index pairs the drivers zip together (truncated to the shorter list)."""
    k = max(int(interval), 1)
    sampled = [n for n in range(n_frames) if n % k == 0]
    following = [n for n in range(n_frames) if (n - 1) % k == 0]
    return sampled, following, list(zip(sampled, following))


def pair_frames(video_frames, framerate):
    """video_frames uint8 [N,H,W,3] (decoded, BGR) -> uint8 [T,2,H,W,3] as relax_fragment_pairs expects."""
    _, _, pairs = sampled_frame_indices(len(video_frames), frame_interval(framerate))
    if not pairs:
        return np.empty((0, 2) + tuple(video_frames.shape[1:]), dtype=video_frames.dtype)
    return np.stack([np.stack([video_frames[a], video_frames[b]]) for a, b in pairs])


def frame_pair_paths(sampled_frame_path, video_name):
    """-> [(frame path, next-frame path), ...] as the reference pairs them (src/main_fragment_layerstack.py:283-293):
    `{video_name}_{n}.png` sorted by n, `{video_name}_{n}_next.png` sorted by n, zipped (the longer list is truncated)."""
    import glob
    base = glob.escape(os.path.join(sampled_frame_path, video_name))
    originals = sorted((p for p in glob.glob(base + "_*.png") if "_next" not in os.path.basename(p)),
                       key=lambda x: int(x.split("_")[-1].split(".")[0]))
    following = sorted(glob.glob(base + "_*_next.png"), key=lambda x: int(x.split("_")[-2]))
    return list(zip(originals, following))


def read_frame_bgr(path):
    """A frame file as `cv2.imread(path)` returns it (src/main_fragment_layerstack.py:295-296): uint8 [H,W,3], BGR, alpha dropped,
    gray replicated.  Decoded with Pillow (the GIL is released while it decodes: loader threads read frames in parallel)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def load_clip_from_frames(sampled_frame_path, video_name, alloc=None):
    """The sampled frames of one video on disk -> uint8 [T,2,H,W,3] BGR, the input of relax_fragment_pairs / of
    dataset.extract_dataset_clips (as its `clips(i)` callable: the decode then runs in the loader threads, ahead of the engine).
    alloc (the dataset driver's protocol): shape -> the uint8 array to fill - pinned staging memory: every frame is decoded and
    written straight into its slot of the clip, no pageable copy of the clip is made.
    Raises if the directory holds no pair or the frames differ in size."""
    pairs = frame_pair_paths(sampled_frame_path, video_name)
    if not pairs:
        raise FileNotFoundError(f"no `{video_name}_<n>.png` / `{video_name}_<n>_next.png` pair under {sampled_frame_path}")
    first = read_frame_bgr(pairs[0][0])
    shape = first.shape
    out = (alloc or np.empty)((len(pairs), 2) + shape) if alloc is not None else np.empty((len(pairs), 2) + shape, dtype=np.uint8)
    for t, (pa, pb) in enumerate(pairs):
        for j, path in enumerate((pa, pb)):
            f = first if (t == 0 and j == 0) else read_frame_bgr(path)
            if f.shape != shape:
                raise ValueError(f"{pa} / {pb}: frame sizes differ inside one video ({f.shape} vs {shape})")
            out[t, j] = f
    return out


def feature_file_name(video_index, network_name, resolution=None):
    """video_index is 0-based like the reference's loop variable i."""
    name = f"{network_name}_feature_map_original" + (f"_{resolution}" if resolution else "")
    return f"video_{video_index + 1}_{name}.npy"


def save_clip_features(directory, video_index, network_name, per_frame_features, skip_existing=False):
    """np.save of the [T,F] array under the reference's file name; skip_existing gives the resume behaviour the
    reference lacks (SURVEY §5 checkpoint/resume)."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, feature_file_name(video_index, network_name))
    if skip_existing and os.path.exists(path):
        return path
    # write under a temporary name, then rename: a run killed mid-write (or a second rank / run reading the directory) never sees
    # a truncated array under the final name (os.replace is atomic within a directory)
    tmp = f"{path}.{os.getpid()}.{threading.get_ident()}.tmp"
    try:
        with open(tmp, "wb") as f:
            np.save(f, np.asarray(per_frame_features))
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return path


def features_matrix(npy_paths):
    """[n_videos, F]: per-video mean over frames (extract_npy2mat.py:117-126)."""
    rows = [np.mean(np.load(p), axis=0) for p in npy_paths]
    out = np.zeros((len(rows),) + rows[0].shape)
    for i, r in enumerate(rows):
        out[i] = r
    return out


def save_mat(path, data_name, matrix):
    import scipy.io
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    scipy.io.savemat(path, {data_name: matrix})
    return path
