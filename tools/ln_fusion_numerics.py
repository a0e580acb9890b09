#!/usr/bin/env python3
"""Would folding LayerNorm into the next GEMM keep fp32-grade results?  (round-2 review, next #7; runs on CPU)

The fusion: the producer epilogue (proj / fc2) emits the residual stream x as split planes plus per-row (sum x, sum x^2); gamma is
folded into the qkv / fc1 weights, W'[n,k] = gamma[k] W[n,k]; the consumer computes
      z[m,n] = rstd[m] * (acc[m,n] - mu[m] * c1[n]) + c2[n],   acc = x W'^T,  c1[n] = sum_k W'[n,k],  c2[n] = beta W^T + b
instead of  z = LN(x) W^T + b.  The difference is pure arithmetic: acc carries the row mean mu through K products and the epilogue
subtracts mu * c1 again - cancellation that grows with |mu| / sigma of the row, and rstd comes from E[x^2] - mu^2.
This script runs both forms in fp32 (numpy float32 matmul, fp32 statistics; the variance of the fused form from the one-pass sums as the
epilogue would have them) on the residual stream of the synthetic ViT-B/16 - regular and adversarial weights - at every block, plus the same
rows shifted by a constant (large-row-offset case), and reports the error of both against float64."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402
from oracle import fragment_ref, vit_ref  # noqa: E402

EPS = 1e-6


def streams(sd, frags, heads=12):
    """the residual stream entering norm1 / norm2 of every block (fp32 oracle forward)"""
    x = vit_ref.preprocess_bgr_u8(frags)
    B = x.shape[0]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    dim, hd = t.shape[-1], t.shape[-1] // heads
    out = []
    for i in range(12):
        p = f"blocks.{i}."
        out.append((p + "norm1", p + "attn.qkv", t.reshape(-1, dim).numpy().copy()))
        y = F.layer_norm(t, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], EPS)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        attn = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
        t = t + F.linear((attn @ qkv[2]).transpose(1, 2).reshape(B, -1, dim), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        out.append((p + "norm2", p + "mlp.fc1", t.reshape(-1, dim).numpy().copy()))
        y = F.gelu(F.linear(F.layer_norm(t, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], EPS), sd[p + "mlp.fc1.weight"],
                            sd[p + "mlp.fc1.bias"]))
        t = t + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return out


def both_forms(x, g, b, W, bias):
    x64, g64, b64, W64, bias64 = (a.astype(np.float64) for a in (x, g, b, W, bias))
    mu64 = x64.mean(1, keepdims=True)
    var64 = x64.var(1, keepdims=True)
    exact = ((x64 - mu64) / np.sqrt(var64 + EPS) * g64 + b64) @ W64.T + bias64
    f = np.float32
    # standard: two-pass LayerNorm in fp32 (csrc/layers.hip), then the GEMM
    mu = x.mean(1, keepdims=True, dtype=f)
    d = x - mu
    var = (d * d).mean(1, keepdims=True, dtype=f)
    y = (d * (f(1) / np.sqrt(var + f(EPS))) * g + b).astype(f)
    std = y @ W.T + bias
    # fused: one-pass row sums (what an epilogue can accumulate), gamma folded into W, correction in the consumer's epilogue
    n = f(x.shape[1])
    s1 = x.sum(1, keepdims=True, dtype=f)
    s2 = (x * x).sum(1, keepdims=True, dtype=f)
    mu_f = s1 / n
    var_f = np.maximum(s2 / n - mu_f * mu_f, f(0))
    rstd = f(1) / np.sqrt(var_f + f(EPS))
    Wp = (W * g[None, :]).astype(f)
    c1 = Wp.sum(1, dtype=f)[None, :]
    c2 = (b[None, :] @ W.T + bias).astype(f)
    acc = x @ Wp.T
    fused = rstd * (acc - mu_f * c1) + c2
    scale = np.abs(exact).mean()
    return np.abs(std - exact).mean() / scale, np.abs(fused - exact).mean() / scale, float(np.abs(mu64).mean() / np.sqrt(var64).mean())


def main():
    frags = []
    for i in range(2):
        o, nx = synth.synthetic_pair(240, 320, 500 + i)
        frags.append(fragment_ref.fragment_pair(o, nx)["ori_frag"])
    frags = np.stack(frags)
    print(f"{'weights':<12}{'site':<22}{'|mu|/sigma':>11}{'standard':>12}{'fused':>12}{'ratio':>8}")
    worst = 0.0
    for adv in (False, True):
        np_sd = synth.vit_state_dict("vit_base", adversarial=adv)
        sd = vit_ref.to_torch_state_dict(np_sd)
        with torch.no_grad():
            sites = streams(sd, frags)
        for ln, lin, x in sites[::3] + [sites[-1]]:
            for shift in (0.0, 30.0):
                e_std, e_fused, ratio_mu = both_forms(x + np.float32(shift), np_sd[ln + ".weight"], np_sd[ln + ".bias"], np_sd[lin + ".weight"],
                                                      np_sd[lin + ".bias"])
                worst = max(worst, e_fused / e_std)
                tag = ("adversarial" if adv else "regular") + (" +30" if shift else "")
                print(f"{tag:<12}{ln:<22}{ratio_mu:>11.2f}{e_std:>12.2e}{e_fused:>12.2e}{e_fused / e_std:>8.1f}")
    print(f"worst fused / standard error ratio: {worst:.1f}")


if __name__ == "__main__":
    main()
