"""One clip through the optical-flow stage a few times (the rocprofv3 target of tools/r04_flow_prof.sh).
   python tools/flow_step.py H W T fused reps"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

H, W, T, fused, reps = (int(a) for a in sys.argv[1:6])
eng = RelaxEngine(0)
eng.set_option("flow_fused", fused)
clip = torch.from_numpy(synth.synthetic_clip(T, H, W, clip_id=5, distinct=2)).cuda()
for _ in range(reps):
    eng.optical_flow(clip, want_flow=False, want_image=True)
torch.cuda.synchronize()
