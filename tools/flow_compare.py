#!/usr/bin/env python3
"""Bitwise regression check of the optical-flow kernels between two builds of the library (GPU box only):
   RELAX_HIP_LIB=old.so python tools/flow_compare.py dump /tmp/a.pt;  python tools/flow_compare.py dump /tmp/b.pt
   python tools/flow_compare.py cmp /tmp/a.pt /tmp/b.pt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smooth_pair(h, w, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.zeros((h, w, 3), np.float32)
    for _ in range(10):
        fx, fy, ph = g.uniform(0.01, 0.12), g.uniform(0.01, 0.12), g.uniform(0, 6.28, 3)
        for c in range(3):
            base[..., c] += g.uniform(10, 30) * np.sin(xx * fx + yy * fy + ph[c])
    a = np.clip(base + 128 + g.normal(0, 2, base.shape), 0, 255).astype(np.uint8)
    b = np.roll(np.roll(a, 2, axis=1), 1, axis=0)
    b[h // 3: 2 * h // 3] = np.roll(a, -3, axis=1)[h // 3: 2 * h // 3]
    return np.stack([a, b])


if sys.argv[1] == "dump":
    import relax_vqa_amd  # noqa: F401
    from relax_vqa_amd.engine import RelaxEngine
    eng = RelaxEngine(0)
    out = {}
    for (h, w, n) in ((200, 264, 2), (270, 483, 1), (540, 960, 3), (1080, 1920, 2), (97, 131, 1)):
        frames = torch.from_numpy(np.stack([smooth_pair(h, w, 7 * i + h) for i in range(n)])).cuda()
        flow, img = eng.optical_flow(frames, want_flow=True, want_image=True)
        out[f"{h}x{w}"] = (flow.cpu(), img.cpu())
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    ok = True
    for k in a:
        fa, ia = a[k]
        fb, ib = b[k]
        same_f, same_i = torch.equal(fa, fb), torch.equal(ia, ib)
        print(f"{k}: flow {'identical' if same_f else 'max abs diff %.3e' % float((fa - fb).abs().max())}, "
              f"image {'identical' if same_i else '%.5f of bytes differ' % float((ia != ib).float().mean())}")
        ok &= same_f and same_i
    print("ALL IDENTICAL" if ok else "DIFFERENT")
