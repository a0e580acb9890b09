#!/usr/bin/env python3
"""Micro-benchmark of the contraction kernel on the shapes of BASELINE config 3 (1 clip = 64 fragments).
   python tools/gemm_bench.py [--iters 20] [--only NAME]      (GPU box only)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd.engine import RelaxEngine, pack_conv_weight  # noqa: E402

ROWS = 64 * 197
GEMMS = {  # name: (M, N, K)
    "vit_qkv": (ROWS, 2304, 768), "vit_proj": (ROWS, 768, 768), "vit_fc1": (ROWS, 3072, 768),
    "vit_fc2": (ROWS, 768, 3072), "vit_patch": (64 * 196, 768, 768),
    "rn_l1_c1": (64 * 3136, 64, 256), "rn_l1_c3": (64 * 3136, 256, 64), "rn_l2_c3": (64 * 784, 512, 128),
    "rn_l3_c1": (64 * 196, 256, 1024), "rn_l3_c3": (64 * 196, 1024, 256), "rn_l4_c1": (64 * 49, 512, 2048),
    "rn_l4_c3": (64 * 49, 2048, 512),
}
CONVS = {  # name: (Nimg, H, Cin, Cout, k, stride, pad)
    "rn_conv1": (64, 224, 4, 64, 7, 2, 3), "rn_l1_c2": (64, 56, 64, 64, 3, 1, 1), "rn_l2_c2": (64, 28, 128, 128, 3, 1, 1),
    "rn_l3_c2": (64, 14, 256, 256, 3, 1, 1), "rn_l4_c2": (64, 7, 512, 512, 3, 1, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"])
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--group-m", type=int, default=0)
    ap.add_argument("--clips", type=int, default=1, help="scale M by this many clips")
    args = ap.parse_args()
    eng = RelaxEngine(0)
    eng.set_precision(args.precision)
    if args.variant >= 0:
        eng.set_option("gemm_variant", args.variant)
    if args.group_m > 0:
        eng.set_option("gemm_group_m", args.group_m)
    dev = torch.device("cuda")
    for name, (M, N, K) in GEMMS.items():
        if args.only and args.only not in name:
            continue
        M *= args.clips
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if args.residual else None
        out = torch.empty(M, N, device=dev)
        for _ in range(3):
            eng.op_gemm(A, W, b, r, act=1, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            eng.op_gemm(A, W, b, r, act=1, out=out)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / args.iters
        print(f"{name:10s} M={M:6d} N={N:4d} K={K:4d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
    for name, (Nimg, H, Cin, Cout, k, stride, pad) in CONVS.items():
        if args.only and args.only not in name:
            continue
        x = torch.randn(Nimg, H, H, Cin, device=dev)
        w = torch.from_numpy(pack_conv_weight(torch.randn(Cout, Cin, k, k).numpy() * (Cin * k * k) ** -0.5)).to(dev)
        b = torch.randn(Cout, device=dev)
        for _ in range(3):
            eng.op_conv2d_nhwc(x, w, b, None, Cout, k, k, stride, pad, 1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            eng.op_conv2d_nhwc(x, w, b, None, Cout, k, k, stride, pad, 1)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / args.iters
        Ho = (H + 2 * pad - k) // stride + 1
        fl = 2.0 * Nimg * Ho * Ho * Cout * Cin * k * k
        print(f"{name:10s} {Nimg}x{H}x{H}x{Cin}->{Cout} k{k}s{stride}  {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s (padded-K flops)", flush=True)


if __name__ == "__main__":
    main()
