#!/usr/bin/env python3
"""Ticks (s_memtime, 100 MHz) per phase of a step of flow_iteration, from the stamped build (GPU box only):
   tools/build_ablations.sh flowstamps;  RELAX_HIP_LIB=tools/abl/librelax_flowstamps.so python tools/flow_stamps.py [H W T]
One producer wave (wave 4) and one box wave (wave 0) of block (1, 1) of pair 0 add their per-phase sums to a device array at the end of
every launch; the table is per step (3 rows of a 240-column band), for the launches of the level whose width is given (0: all levels).
The stamped producer waits for ALL of a step's operands before the first row (the product waits row by row), so `wait for loads` is
an upper bound of the product's exposed wait.  LAB_NOTES.md section 3.3."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import _lib, synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

H, W, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2160, 3840, 2)
lib = ctypes.CDLL(_lib.LIB_PATH)
read = lib.relax_debug_flow_stamps
read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
eng = RelaxEngine(0)
clip = torch.from_numpy(synth.synthetic_clip(T, H, W, clip_id=5, distinct=2)).cuda()
buf = (ctypes.c_ulonglong * 16)()
eng.optical_flow(clip, want_flow=False, want_image=True)
torch.cuda.synchronize()
for level in range(4):
    w = W >> level
    assert read(buf, 1, w) == 0
    eng.optical_flow(clip, want_flow=False, want_image=True)
    torch.cuda.synchronize()
    assert read(buf, 0, 0) == 0
    v = list(buf)
    ps, bs = max(v[2], 1), max(v[7], 1)
    print(f"level {level} ({w} x {H >> level}): {v[2]} producer steps, {v[7]} box steps (three launches)")
    print(f"  producer wave, ticks per step:  wait for loads {v[3] / ps:7.1f}   entries + requests {v[0] / ps:7.1f}   barrier {v[1] / ps:7.1f}"
          f"   sum {(v[0] + v[1] + v[3]) / ps:7.1f}")
    print(f"  box wave, ticks per step:       column sums    {v[4] / bs:7.1f}   strips + solve      {v[5] / bs:7.1f}   barrier {v[6] / bs:7.1f}"
          f"   sum {(v[4] + v[5] + v[6]) / bs:7.1f}")
