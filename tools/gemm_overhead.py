#!/usr/bin/env python3
"""Fit time = rounds * (a * K/BK + b) for the contraction kernel at exact multiples of the resident-workgroup count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa
from relax_vqa_amd.engine import RelaxEngine
eng = RelaxEngine(0)
dev = torch.device("cuda")
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 768
N = 1024
for rounds in (1, 2, 4, 8):
    M = 128 * (slots * rounds // (N // 128))
    for K in (64, 256, 768, 3072):
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; out = torch.empty(M, N, device=dev)
        for _ in range(3): eng.op_gemm(A, W, None, None, act=0, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): eng.op_gemm(A, W, None, None, act=0, out=out)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        print(f"rounds={rounds} M={M} K={K} tiles={M//128*8} {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF  per-round {us/rounds:7.1f} us", flush=True)
