#!/usr/bin/env python3
"""Do two builds of the library give the same bits?  (GPU box only.)
  python tools/ab_bits.py tools/abl/librelax_prev.so relax-vqa_amd/csrc/librelax_hip.so [N] [resnet|vit|both|flow]
Each build runs in its own process (RELAX_HIP_LIB) on the same seeded fragments, for gemm_split_k 0 and 1, with and without the
tap export; prints per output whether the tensors are equal and the largest relative difference."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out, n, what):
    sys.path.insert(0, ROOT)
    import relax_vqa_amd  # noqa: F401
    from relax_vqa_amd import synth
    from relax_vqa_amd.engine import RelaxEngine
    eng = RelaxEngine(0)
    g = torch.Generator().manual_seed(5)
    frags = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, generator=g).cuda()
    res = {}
    if what in ("resnet", "both"):
        eng.load_resnet50(synth.resnet50_state_dict())
        for split in (0, 1):
            eng.set_option("gemm_split_k", split)
            ls, pool = eng.resnet50_features(frags)
            res[f"rn ls split{split}"], res[f"rn pool split{split}"] = ls.cpu(), pool.cpu()
            po = eng.resnet50_features(frags, layer_stack=False)[1]
            res[f"rn pool-only split{split}"] = po.cpu()
            a, b = eng.resnet50_clip_features(frags, n // 2)
            res[f"rn clip ls split{split}"], res[f"rn clip pool split{split}"] = a.cpu(), b.cpu()
        _, _, taps = eng.resnet50_features(frags[:4], taps=range(15))
        for i in range(15):
            res[f"rn tap{i}"] = taps[i].cpu()
    if what == "flow":
        from relax_vqa_amd import synth as sy
        for (hh, ww, t) in ((270, 480, 3), (1080, 1920, 2), (200, 264, 5)):
            fr = torch.stack([torch.stack([torch.from_numpy(a) for a in sy.synthetic_pair(hh, ww, 900 + i)]) for i in range(t)]).cuda()
            fl, im = eng.optical_flow(fr, want_flow=True, want_image=True)
            res[f"flow {hh}x{ww}"], res[f"flow image {hh}x{ww}"] = fl.cpu(), im.cpu().float()
            im_only = eng.optical_flow(fr, want_flow=False, want_image=True)[1]
            res[f"flow image only {hh}x{ww}"] = im_only.cpu().float()
            res[f"flow_to_rgb {hh}x{ww}"] = eng.flow_to_rgb(fl).cpu().float()
    if what in ("vit", "both"):
        eng.load_vit(synth.vit_state_dict("vit_base"), "vit_base")
        for split in (0, 1):
            eng.set_option("gemm_split_k", split)
            res[f"vit split{split}"] = eng.vit_features(frags)[1].cpu()
    torch.save(res, out)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    what = sys.argv[4] if len(sys.argv) > 4 else "resnet"
    outs = []
    for i, lib in enumerate(sys.argv[1:3]):
        out = f"/tmp/ab_bits_{i}.pt"
        env = dict(os.environ, RELAX_HIP_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out, str(n), what], env=env, check=True)
        outs.append(torch.load(out))
    same = True
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        eq = torch.equal(a, b)
        same &= eq
        d = ((a - b).abs().max() / a.abs().max()).item()
        print(f"{k:28s} equal {eq}  max rel diff {d:.2e}")
    print("ALL EQUAL" if same else "DIFFERENT")
