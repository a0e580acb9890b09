#!/usr/bin/env python3
"""Time relax_optical_flow on a synthetic 1080p clip (GPU box only): python tools/flow_bench.py [pairs] [H] [W]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa
from relax_vqa_amd import synth
from relax_vqa_amd.engine import RelaxEngine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
eng = RelaxEngine(0)
clip = torch.from_numpy(synth.synthetic_clip(P, H, W, 1)).cuda()
for _ in range(2): eng.optical_flow(clip)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): eng.optical_flow(clip)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print(f"flow {P} pairs {W}x{H}: {dt*1e3:.1f} ms -> {dt*1e3/P:.3f} ms/pair")
