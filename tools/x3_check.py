#!/usr/bin/env python3
"""bf16x3 tile-variant check: relative error against an fp64 product.   python tools/x3_check.py VARIANT [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

variant = int(sys.argv[1])
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (25001, 768, 768)
eng = RelaxEngine(0)
eng.set_precision("bf16x3")
eng.set_option("gemm_variant", variant)
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda()
r = torch.randn(M, N, generator=g).cuda()
want = torch.relu(A.double() @ W.double().T + b.double() + r.double())
got = eng.op_gemm(A, W, b, r, act=1).double()
print(f"variant {variant} {M}x{N}x{K}: rel err {float((got - want).norm() / want.norm()):.3e}  max abs {float((got - want).abs().max()):.3e}")
