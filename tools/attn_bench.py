#!/usr/bin/env python3
"""Time the fused attention kernel at the config-3 shape (GPU box only): python tools/attn_bench.py [images]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa
from relax_vqa_amd.engine import RelaxEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = RelaxEngine(0)
qkv = torch.randn(n * 197, 2304, device="cuda")
for _ in range(3): eng.op_attention(qkv, n, 12)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): eng.op_attention(qkv, n, 12)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
fl = 4.0 * 197 * 197 * 64 * 12 * n
print(f"attention {n} images x 12 heads: {us:.1f} us  {fl/us/1e6:.1f} TFLOP/s")
