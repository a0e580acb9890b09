#!/usr/bin/env python3
"""Ticks per phase of an (image, head) item of attention_x6, from the stamped build (GPU box only):
   tools/build_ablations.sh a6stamps;  RELAX_HIP_LIB=tools/abl/librelax_a6stamps.so python tools/attn_stamps.py [images]
The stamped kernel writes the per-item averages of workgroup 3's waves 0 and 6 into the first floats of its fp32 output
(wave 0: the older wave of a shared SIMD; wave 6: the younger one of another).  The table of LAB_NOTES.md section 3.2.1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = RelaxEngine(0)
qkv = torch.randn(n * 197, 2304, device="cuda")
out = eng.op_attention(qkv, n, 12)
torch.cuda.synchronize()
v = out.flatten()[:32].cpu().tolist()
names = ["scores", "softmax", "wait for V + barrier", "V rows -> transposed planes + barrier", "P V", "epilogue",
         "split of the next Q", "wait for K + barrier", "K rows -> planes + barrier"]
print(f"{'phase':40s} {'wave 0':>10s} {'wave 6':>10s}")
for i, nm in enumerate(names):
    print(f"{nm:40s} {v[i]:10.0f} {v[16 + i]:10.0f}")
print(f"{'sum':40s} {sum(v[:9]):10.0f} {sum(v[16:25]):10.0f}")
