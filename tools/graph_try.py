#!/usr/bin/env python3
"""Is a step launch-bound?  Captures one whole config-3 step (fragment stage + both backbones) into a HIP graph through
torch.cuda.graph and compares eager and replayed time:  python tools/graph_try.py [clips_per_step]   (GPU box only)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import synth
from relax_vqa_amd.engine import RelaxEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = RelaxEngine(0)
eng.load_resnet50(synth.resnet50_state_dict()); eng.load_vit(synth.vit_state_dict("vit_base"), "vit_base")
clips = [torch.from_numpy(synth.synthetic_clip(32, 1080, 1920, clip_id=i, distinct=4)).cuda() for i in range(B)]
def step():
    return eng.clip_vectors(clips, resnet=True, vit=True)
for _ in range(3): ref = step()
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); e=(time.perf_counter()-t)/10
print(f"eager  B={B}: {e*1e3:.2f} ms/step  {B/e:.2f} clips/s")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph output equal:", torch.equal(out, ref))
t=time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); e=(time.perf_counter()-t)/10
print(f"graph  B={B}: {e*1e3:.2f} ms/step  {B/e:.2f} clips/s")
