"""Does the segmentation of the box-filter rows (where the running column sums restart) change bits of the flow?
   python tools/flow_seg_bits.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402
from tests.test_gpu_flow import _smooth_pair  # noqa: E402

eng = RelaxEngine(0)
for name, frames in (("smooth 540p", np.stack([np.stack(_smooth_pair(540, 960, 3))])), ("noise 1080p", synth.synthetic_clip(2, 1080, 1920, clip_id=5)),
                     ("smooth 1080p", np.stack([np.stack(_smooth_pair(1080, 1920, 4))]))):
    x = torch.from_numpy(frames).cuda()
    ref = None
    for seg in (0, 30, 45, 135, 270):
        eng.set_option("flow_seg_rows", seg)
        fl, _ = eng.optical_flow(x, want_flow=True, want_image=False)
        if ref is None:
            ref = fl.clone()
        else:
            d = (fl - ref).abs()
            print(f"{name}: seg {seg:3d} vs automatic: max |d| {float(d.max()):.3e}, differing {float((fl != ref).float().mean()):.3e}", flush=True)
    eng.set_option("flow_seg_rows", 0)
