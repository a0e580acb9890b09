#!/usr/bin/env python3
"""How close is a build's optical flow to the oracle and to the reference's own OpenCV flow image?  (GPU box only.)
   [RELAX_HIP_LIB=other.so] python tools/flow_margins_ab.py
Prints max / mean |flow - oracle| at 200x264 and 540x960 and the byte identity with tests/golden/png_*/..._residual_of.png:
the numbers the bars of tests/test_gpu_flow.py are set against, for comparing two builds."""
import glob
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relax_vqa_amd  # noqa: E402,F401
from oracle import flow_ref  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402
from tests.test_gpu_flow import _smooth_pair  # noqa: E402

eng = RelaxEngine(0)
for (h, w, seed) in ((200, 264, 1), (540, 960, 3)):
    a, b = _smooth_pair(h, w, seed)
    flow, img = eng.optical_flow(torch.from_numpy(np.stack([a, b])[None]).cuda(), want_flow=True, want_image=True)
    want = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
    err = np.abs(flow[0].cpu().numpy() - want)
    d = np.abs(img[0].cpu().numpy().astype(np.int32) - flow_ref.flow_to_rgb(want).astype(np.int32))
    print(f"{h}x{w}: |flow - oracle| max {err.max():.3e} mean {err.mean():.3e}; image bytes equal {(d == 0).mean():.5f}, within 1 {(d <= 1).mean():.6f}")
for gd in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "png_*"))):
    stem = os.path.basename(gd)[len("png_"):]
    load = lambda suf: np.asarray(Image.open(os.path.join(gd, f"{stem}{suf}.png")).convert("RGB"))[..., ::-1].copy()   # noqa: E731
    if not os.path.exists(os.path.join(gd, f"{stem}_next.png")):
        continue                                     # (the 2160p set holds no frames)
    orig, nxt, want = load(""), load("_next"), load("_residual_of")
    _, img = eng.optical_flow(torch.from_numpy(np.stack([orig, nxt])[None]).cuda())
    d = np.abs(img[0].cpu().numpy().astype(np.int32) - want.astype(np.int32))
    print(f"reference PNG pair {stem} ({orig.shape[1]}x{orig.shape[0]}): bytes equal {(d == 0).mean():.5f}, within 1 {(d <= 1).mean():.6f}, max {d.max()}")
