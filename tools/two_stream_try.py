#!/usr/bin/env python3
"""Do the two backbones overlap usefully on two HIP streams (two handles: one owns ResNet-50, one the ViT)?
   python tools/two_stream_try.py [clips_per_step]      (GPU box only)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa
from relax_vqa_amd import synth
from relax_vqa_amd.engine import RelaxEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e_rn, e_vit = RelaxEngine(0), RelaxEngine(0)
e_rn.load_resnet50(synth.resnet50_state_dict())
e_vit.load_vit(synth.vit_state_dict("vit_base"), "vit_base")
clips = [torch.from_numpy(synth.synthetic_clip(32, 1080, 1920, clip_id=i, distinct=4)).cuda() for i in range(B)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def step(concurrent):
    frs = [e_rn.fragment_pairs(c) for c in clips]
    both = torch.cat([f["ori_frag"] for f in frs] + [f["diff_frag"] for f in frs], dim=0)
    if not concurrent:
        ls, pool = e_rn.resnet50_features(both)
        _, vp = e_vit.vit_features(both, tokens=False, pooled=True)
        return ls, pool, vp
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        ls, pool = e_rn.resnet50_features(both)
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        _, vp = e_vit.vit_features(both, tokens=False, pooled=True)
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)
    return ls, pool, vp


for mode in (False, True, False, True):
    for _ in range(2):
        out = step(mode)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(6):
        out = step(mode)
    torch.cuda.synchronize()
    e = (time.perf_counter() - t) / 6
    print(f"{'two streams' if mode else 'one stream '}  B={B}: {e * 1e3:.2f} ms/step  {B / e:.2f} clips/s")
