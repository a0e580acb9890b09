#!/usr/bin/env python3
"""Fit time = rounds * (a * K/32 + b) for the contraction kernel at exactly 1 and 2 rounds of 512 tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa
from relax_vqa_amd.engine import RelaxEngine
eng = RelaxEngine(0)
dev = torch.device("cuda")
for M in (8192, 16384, 32768):
    for K in (64, 256, 768, 1536, 3072):
        N = 1024
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; out = torch.empty(M, N, device=dev)
        for _ in range(3): eng.op_gemm(A, W, None, None, act=0, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): eng.op_gemm(A, W, None, None, act=0, out=out)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        print(f"M={M} N={N} K={K} tiles={M//128*8} {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF  per-ktile {us/(K/32):6.2f} us", flush=True)
