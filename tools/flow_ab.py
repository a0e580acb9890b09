"""A/B of the optical-flow stage on the GPU: the fused iteration kernel (flow_fused = 1) against update_matrices_k + box_solve_fused
(0): bit equality and time per pair.   python tools/flow_ab.py [H W T]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

H, W, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2160, 3840, 32)
eng = RelaxEngine(0)
clip = torch.from_numpy(synth.synthetic_clip(T, H, W, clip_id=5, distinct=2)).cuda()
res = {}
for mode in (1, 0, 1, 0):
    eng.set_option("flow_fused", mode)
    for _ in range(2):
        fl, im = eng.optical_flow(clip, want_flow=True, want_image=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        eng.optical_flow(clip, want_flow=False, want_image=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n / T * 1e3
    print(f"flow_fused={mode}: {ms:.3f} ms per {W}x{H} pair", flush=True)
    res[mode] = (fl.clone(), im.clone())
print("flow bit-identical:", torch.equal(res[0][0], res[1][0]), " image bit-identical:", torch.equal(res[0][1], res[1][1]),
      " max |d|:", float((res[0][0] - res[1][0]).abs().max()))
