"""Time of the Pillow-exact whole-frame resize (both filters) per clip of T frames.   python tools/resize_time.py H W T"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

H, W, T = (int(a) for a in sys.argv[1:4])
eng = RelaxEngine(0)
frames = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, device="cuda")
for _ in range(2):
    eng.resize_frames(frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    eng.resize_frames(frames)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"resize {T} x {W}x{H}: {ms:.3f} ms  ({T * H * W * 3 / ms / 1e6:.0f} GB/s of input)")
