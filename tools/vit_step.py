#!/usr/bin/env python3
"""One ViT-B/16 pass over N fragments at a chosen precision (GPU box only): the driver for the stamp / ablation builds
(RELAX_HIP_LIB=tools/abl/librelax_x6stamps.so python tools/vit_step.py bf16x6 512) and for quick timing.
A fourth argument "zeros" loads all-zero weights: the same kernels and instruction counts on operands that do not toggle the multipliers -
what the pass would take without the power limit (profiles/r02_micro_mfma_power.txt)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
data = sys.argv[4] if len(sys.argv) > 4 else "random"    # "zeros": all-zero weights (diagnostic: the same instructions on operands that do not toggle the multipliers)
eng = RelaxEngine(0)
sd = synth.vit_state_dict("vit_base")
if data == "zeros":
    sd = {k: v * 0 for k, v in sd.items()}
eng.load_vit(sd, "vit_base")
eng.set_precision(prec)
for kv in os.environ.get("RELAX_OPTS", "").split(","):           # e.g. RELAX_OPTS=h2_kb=0
    if "=" in kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
frags = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device="cuda")
eng.vit_features(frags)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    eng.vit_features(frags)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{prec} ({data} weights): {n} fragments in {dt * 1e3:.1f} ms = {n * 35.126e9 / dt / 1e12:.1f} TFLOP/s algorithmic")
