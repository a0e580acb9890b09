#!/usr/bin/env python3
"""How does v_mfma_f32_32x32x16_bf16 round its 16 products + C?  (GPU box only.)  One K = 16 step through relax_op_gemm
under bf16x6 with operands exactly representable in bf16, so that the (hi, hi) MFMA is the only non-zero one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

eng = RelaxEngine(0)
eng.set_precision("bf16x6")


def run(a_row, w_row):
    A = torch.zeros(64, 16)
    W = torch.zeros(64, 16)
    A[0] = torch.tensor(a_row)
    W[0] = torch.tensor(w_row)
    return eng.op_gemm(A.cuda(), W.cuda())[0, 0].item()


for sign in (+1, -1):
    for e in (25, 26, 27, 28, 29, 30, 34, 40):
        w = [1.0] + [sign * 2.0 ** -e] * 15
        exact = 1.0 + sign * 15 * 2.0 ** -e
        got = run([1.0] * 16, w)
        rne = torch.tensor(exact, dtype=torch.float64).float().item()
        print(f"1 {'+' if sign > 0 else '-'} 15*2^-{e}: exact-1 = {exact - 1:+.6e}  got-1 = {got - 1:+.6e}  fp32-RNE(exact)-1 = {rne - 1:+.6e}")
# two K steps: C enters the second MFMA with a large value, the products are small
for sign in (+1, -1):
    A = torch.zeros(64, 32); W = torch.zeros(64, 32)
    A[0, 0] = 1.0; W[0, 0] = 1.0
    A[0, 16:] = 1.0; W[0, 16:] = sign * 2.0 ** -27
    got = eng.op_gemm(A.cuda(), W.cuda())[0, 0].item()
    print(f"C=1 then 16 products of {'+' if sign > 0 else '-'}2^-27 (sum 1 ulp... exact-1 = {sign * 16 * 2.0 ** -27:+.6e}): got-1 = {got - 1:+.6e}")
