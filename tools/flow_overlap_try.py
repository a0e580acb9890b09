"""Experiment: the input half of the full pipeline (fragments, Farneback flow, resizes: HBM-bound byte work) of step k + 1 on a side
stream under the backbone half (contractions: matrix-core / power-bound) of step k.  Two engines (two handles: separate workspaces) on one
device.   python tools/flow_overlap_try.py [clips_per_step] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W, T = 2160, 3840, 32
front, back = RelaxEngine(0), RelaxEngine(0)
rn, vit = synth.resnet50_state_dict(), synth.vit_state_dict("vit_base")
back.load_resnet50(rn)
back.load_vit(vit, "vit_base")
back.reserve(3 * T * B)
clips = [torch.from_numpy(synth.synthetic_clip(T, H, W, clip_id=50 + i, distinct=2)).cuda() for i in range(2)]
batch = lambda i: [clips[(i * B + j) % 2] for j in range(B)]   # noqa: E731

# sequential: one engine's worth of work per step on one stream (the product path; `back` holds the weights, `front` needs none)
def seq_step(i):
    return back.full_features(front.full_prepare(batch(i), flow=True))

for i in range(2):
    ref = seq_step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    out = seq_step(i)
torch.cuda.synchronize()
t_seq = (time.perf_counter() - t0) / steps
print(f"sequential: {t_seq * 1e3:.1f} ms per step of {B} clips = {B / t_seq:.2f} clips/s", flush=True)

side = torch.cuda.Stream()
main = torch.cuda.current_stream()

def pipelined(n_steps):
    """prepare(k + 1) on the side stream while features(k) run on the main stream; events hand the batches over"""
    outs = []
    with torch.cuda.stream(side):
        prep = front.full_prepare(batch(0), flow=True)
        ready = torch.cuda.Event()
        ready.record(side)
    for k in range(n_steps):
        main.wait_event(ready)
        cur = prep
        if k + 1 < n_steps:
            with torch.cuda.stream(side):
                prep = front.full_prepare(batch(k + 1), flow=True)
                ready = torch.cuda.Event()
                ready.record(side)
        for key in ("rn_in", "vit_in"):
            cur[key].record_stream(main)      # allocated on the side stream, read on the main one
        outs.append(back.full_features(cur))
    return outs

pipelined(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = pipelined(steps)
torch.cuda.synchronize()
t_pipe = (time.perf_counter() - t0) / steps
print(f"pipelined:  {t_pipe * 1e3:.1f} ms per step of {B} clips = {B / t_pipe:.2f} clips/s  ({t_seq / t_pipe:.3f} x)", flush=True)
print("same rows:", bool(torch.equal(outs[(steps - 1)], seq_step(steps - 1))))

# ---- the same pipeline with the two halves on DISJOINT compute units (hipExtStreamCreateWithCUMask): true concurrency ----
import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


for name, w_side, w_main in (("16 of every 32 CUs each", 0x0000FFFF, 0xFFFF0000), ("8 / 24 of every 32", 0x000000FF, 0xFFFFFF00),
                             ("12 / 20 of every 32", 0x00000FFF, 0xFFFFF000)):
    side = masked_stream([w_side] * 8)
    main_m = masked_stream([w_main] * 8)
    main = main_m
    with torch.cuda.stream(main_m):
        pipelined(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = pipelined(steps)
        torch.cuda.synchronize()
        t_m = (time.perf_counter() - t0) / steps
    print(f"CU-masked ({name}): {t_m * 1e3:.1f} ms per step = {B / t_m:.2f} clips/s  ({t_seq / t_m:.3f} x)", flush=True)
