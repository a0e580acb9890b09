"""Soak test of the dataset pass (relax-vqa_amd/dataset.py over the real engine): 300 batches of clips of mixed resolutions and lengths with
a failure injected at every 37th clip - a loader that raises, a malformed clip, an engine call that fails for its whole batch - and per-frame
files written.  The pass runs twice; from the end of the first to the end of the second nothing may grow: the caching allocator's reserved bytes, the free device memory
(hipMemGetInfo: the library's own hipMalloc'd workspaces), the pinned staging pool and its budget book-keeping, the HIP events of the
profiler; and no temporary file stays behind.  Reference: the per-video loop src/main_fragment_layerstack.py:269-361, which leaves its
temporary frame directories behind when a video fails (:330-334); this build must not leak."""
import os

import numpy as np
import pytest
import torch

from relax_vqa_amd import dataset, sampling
from tests.gpu_common import engine, rn50_weights, synth, vit_weights

pytestmark = pytest.mark.gpu

RESOLUTIONS = [(64, 96), (128, 160), (240, 320), (96, 64), (176, 144), (256, 256)]
_cache = {}


def _clip(i):
    h, w = RESOLUTIONS[i % len(RESOLUTIONS)]
    t = 1 + (i // len(RESOLUTIONS)) % 2
    key = (h, w, t)
    if key not in _cache:
        _cache[key] = synth.synthetic_clip(t, h, w, clip_id=900 + len(_cache))
    c = _cache[key].copy()
    c[0, 0, 0, 0, 0] = i % 251                    # (every clip its own bytes; never 255 by accident)
    return c


class FlakyEngine:
    """The real engine, except that a batch containing a marked clip (first byte 255) fails as a whole - what a HIP error would do."""

    def __init__(self, eng):
        self._eng = eng

    def __getattr__(self, name):
        return getattr(self._eng, name)

    def clip_vectors(self, clips, **kw):
        for c in clips:
            if int(c.reshape(-1)[0]) == 255:
                raise RuntimeError("relax_resnet50_clip_features failed (-3): injected fault")
        return self._eng.clip_vectors(clips, **kw)


def _source(i, alloc):
    if i % 37 == 36:
        kind = (i // 37) % 3
        if kind == 0:
            raise OSError(f"video_{i + 1}.mp4: moov atom not found")
        c = _clip(i)
        if kind == 1:
            return c[:, :1]                          # malformed: no `next` frame (refused after the decode)
        c[0, 0, 0, 0, 0] = 255                       # the engine fails on this one (and on the batch it is in: the driver retries clip by clip)
        out = alloc(c.shape)
        np.copyto(out, c)
        return out
    c = _clip(i)
    out = alloc(c.shape)                             # decode straight into the pinned staging pool (the `alloc` protocol)
    np.copyto(out, c)
    return out


def _state(eng, stager):
    torch.cuda.synchronize()
    return {"reserved": torch.cuda.memory_reserved(), "free": torch.cuda.mem_get_info()[0], "workspace_mib": eng.get_option("workspace_mib"),
            "events": eng.get_option("profile_events"), "pinned_live": stager.pinned_live_bytes, "in_use": stager.gate.in_use,
            "landing": len(stager._landing)}


def test_soak_300_batches_of_mixed_resolutions_with_injected_failures(tmp_path):
    rn50_weights(), vit_weights("vit_base")
    real = engine()
    eng = FlakyEngine(real)
    out_dir = str(tmp_path / "features")
    real.profile_enable(True)                        # the event pool is exercised too: spans are folded into the totals as they pile up
    n = 600                                          # 300 batches of 2
    try:
        # the same pass twice: after the first one workspaces, result matrices, pinned pool and event pool have reached their size
        m0, e0 = dataset.extract_dataset_clips(_source, n, eng, clips_per_step=2, out_dir=out_dir, rank=0, world=1, prefetch=2, workers=4)
        real.profile_read(0)                         # drains the spans: their events go back to the pool
        stager = dataset._stager(eng)
        m0 = m0.cpu()                                # (the first pass's matrix leaves the device: the second one's takes its block)
        warm = _state(real, stager)
        matrix, errors = dataset.extract_dataset_clips(_source, n, eng, clips_per_step=2, out_dir=out_dir, rank=0, world=1, prefetch=2, workers=4)
        real.profile_read(0)
        matrix = matrix.cpu()
        after = _state(real, stager)
    finally:
        real.profile_enable(False)
    bad = [i for i in range(n) if i % 37 == 36]
    assert [i for i, _ in errors] == bad and len(bad) == 16
    assert bool(torch.isnan(matrix[bad]).all()) and not bool(torch.isnan(matrix[[i for i in range(n) if i % 37 != 36]]).any())
    ok = ~torch.isnan(m0).any(dim=1)
    assert torch.equal(matrix[ok], m0[ok]), "a clip's row changed between two passes"
    print(f"\nsoak: warm-up {warm}\n      after   {after}")
    assert after["reserved"] <= warm["reserved"], "the caching allocator grew during the soak"
    assert after["free"] >= warm["free"] - (8 << 20), "device memory outside the caching allocator was lost"
    assert after["workspace_mib"] == warm["workspace_mib"]
    assert after["events"] <= warm["events"] + 64 and after["events"] <= 3 * 2 * 2048, "HIP events leaked"   # (bounded: ~2 x kReapAt spans, not 600 clips x ~250 launches)
    assert after["pinned_live"] <= max(warm["pinned_live"], stager.pool_limit_bytes) and after["in_use"] == 0 and after["landing"] == 0
    files = sorted(os.listdir(out_dir))
    assert len(files) == n - len(bad) and all(f.endswith(".npy") and ".tmp" not in f for f in files), "temporary or stray files were left behind"
    assert sampling.feature_file_name(0, "resnet50") in files
