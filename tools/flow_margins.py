import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import relax_vqa_amd  # noqa: F401
from oracle import flow_ref, fragment_ref
from tests.gpu_common import engine
from tests.test_gpu_flow import _smooth_pair, _load
gd = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
orig, nxt, want = _load(gd, ""), _load(gd, "_next"), _load(gd, "_residual_of")
_, img = engine().optical_flow(torch.from_numpy(np.stack([orig, nxt])[None]).cuda())
d = np.abs(img[0].cpu().numpy().astype(np.int32) - want.astype(np.int32))
print("png pair: exact", (d == 0).mean(), "(bar 0.999)  <=1", (d <= 1).mean(), "(bar 0.9999)")
fo = engine().fragment_image(img)
n = int(fo["counts"][0]); got = set(map(tuple, fo["positions"][0, :n].cpu().numpy().tolist()))
_, wp = fragment_ref.extract_important_patches(want, fragment_ref.get_patch_diff(want))
print("positions matched", len(got & set(map(tuple, wp.tolist()))), "(bar 195)")
for (h, w, s) in ((200, 264, 1), (97, 131, 5), (150, 203, 5), (136, 240, 5)):
    a, b = _smooth_pair(h, w, s)
    flow, im = engine().optical_flow(torch.from_numpy(np.stack([a, b])[None]).cuda(), want_flow=True, want_image=True)
    wf = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
    err = np.abs(flow[0].cpu().numpy() - wf)
    wi = flow_ref.flow_to_rgb(wf)
    dd = np.abs(im[0].cpu().numpy().astype(np.int32) - wi.astype(np.int32))
    print(f"{h}x{w}: flow err max {err.max():.2e} (bar 2e-2) mean {err.mean():.2e} (bar 1e-4); image exact {(dd==0).mean():.5f} (bar 0.995) <=1 {(dd<=1).mean():.5f} (bar 0.9995)")
fr = engine().fragment_pairs(torch.from_numpy(np.stack([orig, nxt])[None]).cuda())
merged = engine().merge_fragments(fr["diff_frag"], fo["frag"])[0].cpu().numpy()
print("merged fragment vs reference PNG: exact", (merged == _load(gd, "_residual_merged_frag")).mean(), "(bar 0.995)")
