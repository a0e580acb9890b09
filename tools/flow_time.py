"""Time per pair of the optical-flow stage with whatever library RELAX_HIP_LIB names (ablation builds: tools/build_ablations.sh flow:<mask>).
   python tools/flow_time.py H W T [label]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

H, W, T = (int(a) for a in sys.argv[1:4])
label = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(os.environ.get("RELAX_HIP_LIB", "product")) + " " + os.environ.get("RELAX_FLOW_OPTS", "")
eng = RelaxEngine(0)
for kv in os.environ.get("RELAX_FLOW_OPTS", "").split(","):      # e.g. RELAX_FLOW_OPTS=flow_seg_rows=540,flow_fused=0
    if "=" in kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
clip = torch.from_numpy(synth.synthetic_clip(T, H, W, clip_id=5, distinct=2)).cuda()
for _ in range(2):
    eng.optical_flow(clip, want_flow=False, want_image=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for _ in range(n):
    eng.optical_flow(clip, want_flow=False, want_image=True)
torch.cuda.synchronize()
print(f"{label:28s} {(time.perf_counter() - t0) / n / T * 1e3:.3f} ms per {W}x{H} pair", flush=True)
