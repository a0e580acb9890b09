import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import relax_vqa_amd
from relax_vqa_amd import synth
from oracle import vit_ref
torch.set_num_threads(8)
LN_EPS = vit_ref.LN_EPS

def trunc22(x, s):   # two fp16 planes of x*s
    xs = x.double() * s
    hi = xs.to(torch.float16).double()
    lo = (xs - hi).to(torch.float16).double()
    return (hi + lo) / s

def pow2_for(bound):
    e = np.floor(np.log2(bound)) + 1
    return float(2.0 ** (15 - e))

def forward(sd, x, heads, mode, loose=8.0):
    B = x.shape[0]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    dim = t.shape[-1]; hd = dim // heads
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p = f"blocks.{i}."
        y = F.layer_norm(t, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if mode == "f64":
            attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
            o = attn @ v
        elif mode == "f32":
            q32, k32, v32 = q.float(), k.float(), v.float()
            attn = ((q32 @ k32.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
            o = (attn @ v32).double()
        else:  # h2: 22-bit operands, static scale from a loose bound on the whole qkv tensor
            s = pow2_for(loose * float(qkv.abs().max()))
            q2, k2, v2 = trunc22(q.float(), s), trunc22(k.float(), s), trunc22(v.float(), s)
            sc = (q2 @ k2.transpose(-2, -1)).float() * np.float32(hd ** -0.5 * 1.4426950408889634)   # fp32 accumulator, log2e folded
            mx = sc.max(dim=-1, keepdim=True).values
            e = torch.exp2(sc - mx)                    # fp32
            ssum = e.sum(dim=-1, keepdim=True)
            p2 = trunc22(e, 2.0 ** 14)
            o = ((p2 @ v2).float() / ssum).double()
        y = o.transpose(1, 2).reshape(B, -1, dim)
        t = t + F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y = F.layer_norm(t, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        t = t + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    t = F.layer_norm(t, (dim,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return t[:, 1:]

rng = np.random.default_rng(5)
frags = rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8)
x = vit_ref.preprocess_bgr_u8(frags).double()
for adv in (False, True):
    sd = {k: torch.from_numpy(v).double() for k, v in synth.vit_state_dict("vit_base", adversarial=adv).items()}
    ref = forward(sd, x, 12, "f64")
    for mode, loose in (("f32", 0), ("h2", 2.0), ("h2", 64.0), ("h2", 4096.0)):
        got = forward(sd, x, 12, mode, loose)
        print("adv" if adv else "reg", mode, loose, "norm-rel err of the tokens (attention arithmetic alone):", float((got - ref).norm() / ref.norm()))
