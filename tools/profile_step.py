#!/usr/bin/env python3
"""A few config-3 steps in one precision, for rocprofv3:  rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/profile_step.py bf16x3"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
clips_per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = RelaxEngine(0)
eng.load_resnet50(synth.resnet50_state_dict())
eng.load_vit(synth.vit_state_dict("vit_base"), "vit_base")
eng.set_precision(prec)
clip = torch.from_numpy(synth.synthetic_clip(32, 1080, 1920, clip_id=0, distinct=4)).cuda()
clips = [clip] * clips_per_step
for _ in range(3):
    eng.clip_vectors(clips, resnet=True, vit=True)
torch.cuda.synchronize()
