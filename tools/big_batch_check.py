#!/usr/bin/env python3
"""Index-overflow check at batch sizes far beyond the bench's (GPU box only): 64 and 96 clips of 1080p x 32 pairs in ONE pass
(4096 / 6144 fragments per backbone pass, activations of tens of GB) must give bit-identical rows to a 2-clip pass with the tail
split-K off; same for the full 35203-d vectors with 12 clips per pass.  Measured: identical."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import relax_vqa_amd  # noqa: F401  (registers the package alias)
from relax_vqa_amd import synth
from relax_vqa_amd.engine import RelaxEngine
eng = RelaxEngine(0)
eng.load_resnet50(synth.resnet50_state_dict()); eng.load_vit(synth.vit_state_dict("vit_base"), "vit_base")
clips = [torch.from_numpy(synth.synthetic_clip(32, 1080, 1920, clip_id=i, distinct=4)).cuda() for i in range(2)]
eng.set_option("gemm_split_k", 0)
ref = eng.clip_vectors(clips)
for B in (64, 96):
    out = eng.clip_vectors([clips[i % 2] for i in range(B)])
    ok = all(torch.equal(out[i], ref[i % 2]) for i in range(B))
    print(B, "clips per pass:", "bit-identical to the 2-clip pass" if ok else "DIFFERENT", float((out - ref[[i % 2 for i in range(B)]]).abs().max()))
full = eng.full_clip_vectors(clips[:1], flow=True)
outf = eng.full_clip_vectors([clips[0]] * 12, flow=True)
print("full 12:", all(torch.equal(outf[i], full[0]) for i in range(12)))
