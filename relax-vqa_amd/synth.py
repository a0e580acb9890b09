"""Deterministic synthetic weights and clips (no network: no checkpoints, no videos).

Weights follow the reference's state-dict key names so a real torchvision
``resnet50`` / DINO ``vitbase16`` checkpoint drops into the same loader:
  ResNet-50 keys: torchvision (reference: src/extractor/visualise_resnet.py:21)
  ViT keys:       DINO checkpoint loaded at src/extractor/visualise_vit_layer.py:326-328
Everything is drawn from numpy PCG64 so host, oracle and GPU box regenerate
bit-identical tensors without shipping ~440 MB.
"""
import numpy as np

RESNET_STAGES = [(1, 3, 64, 1), (2, 4, 128, 2), (3, 6, 256, 2), (4, 3, 512, 2)]


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def resnet50_state_dict(seed=7, adversarial=False):
    """fp32 numpy state dict with torchvision resnet50 key names (fc omitted:
    the reference computes fc but never reads it).
    adversarial=True: a second weight set built to stress the BatchNorm / ReLU arithmetic - running variances log-uniform over
    [1e-3, 10] (eps = 1e-5 matters at the low end; the producing conv's rows are scaled by sqrt(var) so activations stay O(1)),
    gammas of mixed sign (ReLU keeps the other half), larger running means and biases.
    adversarial="outliers": a third set with the pathology of real checkpoints that the per-image Hoelder scales of the f16x2
    layers must survive - in every stage ONE output channel of the first block (bn3 and downsample BatchNorm gammas x 60, positive
    shift) is 50 - 100 x the others and rides the identity path through the stage, so max |x| of an image is set by one channel and
    every other value of the tensor sits 6 - 7 binades below the scale's top; and a group of 16 channels per block's bn1 / bn2 is
    dead (gamma ~ 0, shift -10: all-zero behind the ReLU)."""
    outliers = isinstance(adversarial, str) and adversarial == "outliers"
    if outliers:
        adversarial = False
    g = _rng(seed + (1000 if adversarial else 0) + (2000 if outliers else 0))
    sd = {}
    last_conv = [None]

    def conv(name, cout, cin, k):
        fan_out = cout * k * k
        sd[name + ".weight"] = (g.standard_normal((cout, cin, k, k), dtype=np.float32)
                                * np.float32(np.sqrt(2.0 / fan_out)))
        last_conv[0] = name + ".weight"

    def bn(name, c, gamma_scale=1.0):
        if adversarial:
            var = np.exp(g.uniform(np.log(1e-3), np.log(10.0), c))
            sd[name + ".weight"] = (g.uniform(0.5, 1.5, c) * g.choice([-1.0, 1.0], c) * gamma_scale).astype(np.float32)
            sd[name + ".bias"] = (g.standard_normal(c) * 0.3).astype(np.float32)
            sd[name + ".running_mean"] = (g.standard_normal(c) * 0.5 * np.sqrt(var)).astype(np.float32)
            sd[name + ".running_var"] = var.astype(np.float32)
            sd[last_conv[0]] = (sd[last_conv[0]] * np.sqrt(var).astype(np.float32)[:, None, None, None]).astype(np.float32)
            return
        sd[name + ".weight"] = (g.uniform(0.5, 1.5, c) * gamma_scale).astype(np.float32)
        sd[name + ".bias"] = (g.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (g.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = g.uniform(0.5, 1.5, c).astype(np.float32)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    cin = 64
    for layer, blocks, width, _stride in RESNET_STAGES:
        for b in range(blocks):
            p = f"layer{layer}.{b}"
            conv(p + ".conv1", width, cin, 1)
            bn(p + ".bn1", width)
            conv(p + ".conv2", width, width, 3)
            bn(p + ".bn2", width)
            conv(p + ".conv3", width * 4, width, 1)
            bn(p + ".bn3", width * 4, gamma_scale=0.5)   # keeps the residual sum from growing 2x per block
            if b == 0:
                conv(p + ".downsample.0", width * 4, cin, 1)
                bn(p + ".downsample.1", width * 4)
            cin = width * 4
    if outliers:
        for layer, blocks, width, _stride in RESNET_STAGES:
            hot = int(g.integers(0, width * 4))
            for name in (f"layer{layer}.0.bn3", f"layer{layer}.0.downsample.1"):
                sd[name + ".weight"][hot] = np.float32(abs(sd[name + ".weight"][hot]) * (240.0 if layer == 4 else 60.0))
                sd[name + ".bias"][hot] = np.float32(4.0)
            for b in range(blocks):
                for bnk, c in ((f"layer{layer}.{b}.bn1", width), (f"layer{layer}.{b}.bn2", width)):
                    d0 = int(g.integers(0, c // 16)) * 16
                    sd[bnk + ".weight"][d0:d0 + 16] = np.float32(1e-3)
                    sd[bnk + ".bias"][d0:d0 + 16] = np.float32(-10.0)
    return sd


def vit_state_dict(name_model="vit_base", patch=16, seed=11, adversarial=False):
    """fp32 numpy state dict with DINO ViT key names.  Biases / LN affine are
    non-trivial on purpose so a missing bias add cannot pass parity.
    adversarial=True: a second weight set for the softmax path - the q and k rows of every qkv matrix are scaled so attention
    logits span about +-20 (peaked, near one-hot rows next to flat ones: the max subtraction and the exp2 of the attention
    kernel work at their limits), LayerNorm gammas of mixed sign.
    adversarial="outliers": a third set with the pathology of real DINO / ViT checkpoints that the STATIC scales of the f16x2 path
    (bounds over every possible input: csrc/h2.h, host_logic.cpp) must survive - five residual-stream channels carry values hundreds
    of times the median (pos_embed of +-6 there, fc2 / proj rows x 20 and biases of +-3 writing into them in every block), the
    LayerNorm gains and shifts on those channels go up to 10 and 5 (so the Cauchy-Schwarz bound behind a LayerNorm is set by five
    of 768 channels and is 2^7 .. 2^10 loose for all the others), and head 0 of every block has its q / k rows x 4 (near-one-hot
    softmax rows next to ordinary heads)."""
    cfg = {"vit_tiny": (192, 12, 3), "vit_small": (384, 12, 6), "vit_base": (768, 12, 12)}[name_model]
    dim, depth, _heads = cfg
    outliers = isinstance(adversarial, str) and adversarial == "outliers"
    if outliers:
        adversarial = False
    g = _rng(seed + (1000 if adversarial else 0) + (2000 if outliers else 0))

    def nrm(shape, std):
        return (g.standard_normal(shape, dtype=np.float32) * np.float32(std))

    sd = {
        "cls_token": nrm((1, 1, dim), 0.02),
        "pos_embed": nrm((1, 197, dim), 0.02),
        "patch_embed.proj.weight": nrm((dim, 3, patch, patch), 0.02),
        "patch_embed.proj.bias": nrm((dim,), 0.02),
    }
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = g.uniform(0.5, 1.5, dim).astype(np.float32)
        sd[p + "norm1.bias"] = nrm((dim,), 0.1)
        sd[p + "attn.qkv.weight"] = nrm((3 * dim, dim), 0.05)
        sd[p + "attn.qkv.bias"] = nrm((3 * dim,), 0.02)
        if adversarial:
            sd[p + "norm1.weight"] *= g.choice([-1.0, 1.0], dim).astype(np.float32)
            sd[p + "attn.qkv.weight"][: 2 * dim] *= np.float32(3.0)      # q and k rows: logits (q.k / 8) of std ~18
        sd[p + "attn.proj.weight"] = nrm((dim, dim), 0.02)
        sd[p + "attn.proj.bias"] = nrm((dim,), 0.02)
        sd[p + "norm2.weight"] = g.uniform(0.5, 1.5, dim).astype(np.float32)
        sd[p + "norm2.bias"] = nrm((dim,), 0.1)
        sd[p + "mlp.fc1.weight"] = nrm((4 * dim, dim), 0.02)
        sd[p + "mlp.fc1.bias"] = nrm((4 * dim,), 0.02)
        sd[p + "mlp.fc2.weight"] = nrm((dim, 4 * dim), 0.02)
        sd[p + "mlp.fc2.bias"] = nrm((dim,), 0.02)
    sd["norm.weight"] = g.uniform(0.5, 1.5, dim).astype(np.float32)
    sd["norm.bias"] = nrm((dim,), 0.1)
    if outliers:
        hot = np.sort(g.choice(dim, 5, replace=False))
        hd = dim // _heads
        # (a DC component per channel, the same sign for every token - what real checkpoints show - plus 10 % token-to-token variation)
        sd["pos_embed"][..., hot] = (g.choice([-1.0, 1.0], (1, 1, 5)) * g.uniform(150.0, 250.0, (1, 197, 5))).astype(np.float32)
        for i in range(depth):
            p = f"blocks.{i}."
            for w, b in (("attn.proj.weight", "attn.proj.bias"), ("mlp.fc2.weight", "mlp.fc2.bias")):
                sd[p + w][hot] *= np.float32(20.0)
                sd[p + b][hot] = (g.choice([-1.0, 1.0], 5) * g.uniform(5.0, 15.0, 5)).astype(np.float32)
            for nk in ("norm1", "norm2"):
                sd[p + nk + ".weight"][hot] = g.uniform(4.0, 10.0, 5).astype(np.float32)
                sd[p + nk + ".bias"][hot] = (g.choice([-1.0, 1.0], 5) * g.uniform(2.0, 5.0, 5)).astype(np.float32)
            # (the consumers read those channels with small weights, as trained networks do: the stream's other channels stay O(1))
            sd[p + "attn.qkv.weight"][:, hot] *= np.float32(0.1)
            sd[p + "mlp.fc1.weight"][:, hot] *= np.float32(0.1)
            qkv = sd[p + "attn.qkv.weight"]
            qkv[0:hd] *= np.float32(12.0)                   # q rows of head 0
            qkv[dim:dim + hd] *= np.float32(12.0)           # k rows of head 0
        sd["norm.weight"][hot] = g.uniform(4.0, 10.0, 5).astype(np.float32)
        sd["norm.bias"][hot] = (g.choice([-1.0, 1.0], 5) * g.uniform(2.0, 5.0, 5)).astype(np.float32)
    return sd


def mlp_head_state_dict(input_features=35203, hidden=256, seed=23):
    """fp32 numpy state dict with the key names of the reference's Mlp (src/model_regression.py:37-58)."""
    g = _rng(seed)

    def nrm(shape, std):
        return g.standard_normal(shape, dtype=np.float32) * np.float32(std)

    return {
        "fc1.weight": nrm((hidden, input_features), input_features ** -0.5 * 4.0),
        "fc1.bias": nrm((hidden,), 0.1),
        "bn1.weight": g.uniform(0.5, 1.5, hidden).astype(np.float32),
        "bn1.bias": nrm((hidden,), 0.1),
        "bn1.running_mean": nrm((hidden,), 0.5),
        "bn1.running_var": g.uniform(0.5, 2.0, hidden).astype(np.float32),
        "fc2.weight": nrm((hidden // 2, hidden), hidden ** -0.5 * 2.0),
        "fc2.bias": nrm((hidden // 2,), 0.1),
        "fc3.weight": nrm((1, hidden // 2), (hidden // 2) ** -0.5 * 20.0),
        "fc3.bias": np.float32([50.0]),
    }


def synthetic_pair(height, width, seed, patch=16, max_amp=64):
    """One (orig, next) pair, uint8 [H,W,3] BGR each.  next = clip(orig + noise)
    with a per-16x16-patch noise amplitude in {0..max_amp}, so patch scores are
    spread the way sparse motion spreads them."""
    g = _rng(seed)
    orig = g.integers(0, 256, (height, width, 3), dtype=np.uint8)
    ph, pw = -(-height // patch), -(-width // patch)
    amp = g.integers(0, max_amp + 1, (ph, pw), dtype=np.int16)
    amp_px = np.repeat(np.repeat(amp, patch, axis=0), patch, axis=1)[:height, :width, None]
    noise = g.integers(-max_amp, max_amp + 1, (height, width, 3), dtype=np.int16)
    noise = np.sign(noise) * np.minimum(np.abs(noise), amp_px)
    nxt = np.clip(orig.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return orig, nxt


def synthetic_clip(n_pairs, height, width, clip_id=0, distinct=None):
    """uint8 [T,2,H,W,3]: frames[t,0]=sampled frame, frames[t,1]=the frame after
    it (the pairing of src/video_frames_extract.py:51-69).  `distinct` < n_pairs draws only that many pairs from the
    generator and derives the rest by rolling them by whole patches (different content and index maps per pair at a
    fraction of the host time; used by bench.py, never by the parity tests)."""
    out = np.empty((n_pairs, 2, height, width, 3), dtype=np.uint8)
    distinct = n_pairs if distinct is None else max(1, min(distinct, n_pairs))
    for t in range(distinct):
        o, n = synthetic_pair(height, width, seed=1000 + clip_id * 4096 + t)
        out[t, 0], out[t, 1] = o, n
    for t in range(distinct, n_pairs):
        shift = 16 * (t // distinct)
        out[t] = np.roll(out[t % distinct], shift=(shift, 3 * shift), axis=(1, 2))
    return out
