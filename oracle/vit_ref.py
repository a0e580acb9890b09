"""Oracle (TEST INFRASTRUCTURE, not product): CPU fp32 restatement of the DINO
ViT forward used by the reference extractor.

Follows /root/reference/src/extractor/visualise_vit_layer.py:
  :132-149  PatchEmbed  (conv k=p, stride=p, flatten, transpose)
  :221-232  prepare_tokens (cls prepend, + pos_embed; no interpolation at 224^2, :197-201)
  :93-106   Attention (qkv -> [3,B,h,N,d]; softmax(q k^T * d^-.5) v; proj)
  :123-129  Block (pre-LN, residual twice)
  :62-78    Mlp (fc1, exact-erf GELU, fc2)
  :234-239  forward: final LayerNorm, returns x[:,0], x[:,1:]
  :287-289  vit_base: dim 768, depth 12, heads 12, mlp 4x, qkv_bias, LN eps 1e-6
  :339-342,466-470,492-494  input: PIL RGB, /255, NO mean/std normalisation
Pinned against the reference class itself by oracle/make_golden.py.
State-dict keys are the DINO checkpoint's (cls_token, pos_embed,
patch_embed.proj.weight, blocks.0.attn.qkv.weight, ..., norm.weight).
"""
import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-6

VIT_CONFIGS = {
    "vit_tiny": dict(dim=192, depth=12, heads=3),
    "vit_small": dict(dim=384, depth=12, heads=6),
    "vit_base": dict(dim=768, depth=12, heads=12),
}


def preprocess_bgr_u8(frag_bgr_u8):
    """uint8 [N,224,224,3] BGR -> fp32 [N,3,224,224] RGB in [0,1]."""
    x = torch.as_tensor(np.ascontiguousarray(frag_bgr_u8[..., ::-1]))
    return x.permute(0, 3, 1, 2).to(torch.float32).div(255)


@torch.no_grad()
def forward_tokens(sd, x, heads, patch=16):
    """x fp32 [B,3,224,224] -> fp32 [B,196,dim] final-norm patch tokens."""
    B = x.shape[0]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    dim = t.shape[-1]
    hd = dim // heads
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p = f"blocks.{i}."
        y = F.layer_norm(t, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
        y = (attn @ v).transpose(1, 2).reshape(B, -1, dim)
        t = t + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y = F.layer_norm(t, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        t = t + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    t = F.layer_norm(t, (dim,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return t[:, 1:]


def to_torch_state_dict(np_sd):
    return {k: torch.as_tensor(np.asarray(v)) for k, v in np_sd.items()}


def tokens(sd, frag_bgr_u8, heads=12):
    return forward_tokens(sd, preprocess_bgr_u8(frag_bgr_u8), heads).numpy()


def pool_features(sd, frag_bgr_u8, heads=12):
    """uint8 BGR fragments [N,224,224,3] -> fp32 [N, 3*dim] (mean|max|std over tokens)."""
    t = tokens(sd, frag_bgr_u8, heads)
    return np.concatenate([t.mean(axis=1), t.max(axis=1), t.std(axis=1)], axis=1).astype(np.float32)
