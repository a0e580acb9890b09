#!/usr/bin/env python3
"""One ResNet-50 pass over N fragments (GPU box only): the quick A/B target for the convolution kernels
(RELAX_HIP_LIB=... python tools/resnet_step.py 1024 5 [ls|pool|both]) and the rocprofv3 target of tools/resnet_layers.sh."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "both"
eng = RelaxEngine(0)
eng.load_resnet50(synth.resnet50_state_dict())
for kv in os.environ.get("RELAX_OPTS", "").split(","):           # e.g. RELAX_OPTS=x6_narrow_k=512
    if "=" in kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
frags = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device="cuda")
kw = dict(layer_stack=what in ("ls", "both"), pool=what in ("pool", "both"))
eng.resnet50_features(frags, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    eng.resnet50_features(frags, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(os.environ.get("RELAX_OPTS", ""), end=" ")
print(f"resnet50 ({what}): {n} fragments in {dt * 1e3:.2f} ms = {n * 8.174e9 / dt / 1e12:.1f} TFLOP/s algorithmic, "
      f"{6 * n * 8.174e9 / dt / 2.5e15:.3f} of the dense bf16 peak executed")
