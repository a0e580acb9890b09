"""Run INSIDE a subprocess with libasan preloaded (tests/test_host_logic_sanitized.py): drives the pure-host half of
librelax_hip.so (csrc/host_logic.cpp built with -fsanitize=address,undefined) with the synthetic state dicts and with the
deliberately malformed ones of test_errors_are_reported_not_swallowed.  Any sanitizer report aborts the process."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import synth  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "relax-vqa_amd", "csrc", "librelax_host_san.so"))
lib.relax_host_pack_conv.restype = C.c_int
lib.relax_host_conv_kpad.restype = C.c_int
lib.relax_host_fold_fc_bn.restype = C.c_int


def marshal(sd):
    names = [k.encode() for k in sd]
    arrays = [np.ascontiguousarray(v, dtype=np.float32) for v in sd.values()]
    n = len(names)
    return ((C.c_void_p * n)(*[a.ctypes.data for a in arrays]), (C.c_char_p * n)(*names), (C.c_int64 * n)(*[a.size for a in arrays]), n,
            arrays)


def pack(sd, conv, bn, cout, cin, cin_pad, k):
    ptrs, names, numels, n, keep = marshal(sd)
    kpad = lib.relax_host_conv_kpad(k, cin_pad)
    out = np.full((cout, kpad), np.nan, dtype=np.float32)          # exact size: an overrun is an ASan report
    shift = np.full((cout,), np.nan, dtype=np.float32)
    err = C.create_string_buffer(256)
    rc = lib.relax_host_pack_conv(ptrs, names, numels, n, conv.encode(), bn.encode(), cout, cin, cin_pad, k,
                                  out.ctypes.data_as(C.c_void_p), shift.ctypes.data_as(C.c_void_p), err, 256)
    return rc, out, shift, err.value.decode()


def want_pack(sd, conv, bn, cin_pad):
    w = sd[conv + ".weight"]
    cout, cin, k, _ = w.shape
    scale = np.ones(cout, np.float32)
    shift = None
    if bn:
        scale = (sd[bn + ".weight"] / np.sqrt(sd[bn + ".running_var"] + np.float32(1e-5))).astype(np.float32)
        shift = (sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale).astype(np.float32)
    kpad = -(-(k * k * cin_pad) // 32) * 32
    t = np.zeros((cout, k, k, cin_pad), np.float32)
    t[..., :cin] = (w * scale[:, None, None, None]).transpose(0, 2, 3, 1)
    out = np.zeros((cout, kpad), np.float32)
    out[:, :k * k * cin_pad] = t.reshape(cout, -1)
    return out, shift


rn = synth.resnet50_state_dict()
# every convolution of the network, its BatchNorm folded in (conv1: raw, Cin padded 3 -> 4, K 196 -> 224)
cases = [("conv1", "", 64, 3, 4, 7)]
cin = 64
for layer, blocks, width, _ in synth.RESNET_STAGES:
    for b in range(blocks):
        p = f"layer{layer}.{b}"
        cases += [(p + ".conv1", p + ".bn1", width, cin, cin, 1), (p + ".conv2", p + ".bn2", width, width, width, 3),
                  (p + ".conv3", p + ".bn3", width * 4, width, width, 1)]
        if b == 0:
            cases.append((p + ".downsample.0", p + ".downsample.1", width * 4, cin, cin, 1))
        cin = width * 4
for conv, bn, cout, ci, cp, k in cases:
    rc, out, shift, err = pack(rn, conv, bn, cout, ci, cp, k)
    assert rc == 0, (conv, err)
    w_out, w_shift = want_pack(rn, conv, bn, cp)
    assert np.array_equal(out, w_out), conv
    if bn:
        assert np.array_equal(shift, w_shift), conv
print(f"packed {len(cases)} convolutions, bit-identical to the numpy restatement")

# malformed state dicts: an error string, never a read past a buffer
bad = dict(rn)
del bad["layer3.4.conv2.weight"]
rc, _, _, err = pack(bad, "layer3.4.conv2", "layer3.4.bn2", 256, 256, 256, 3)
assert rc == -1 and "missing key 'layer3.4.conv2.weight'" in err, err
bad = dict(rn)
bad["conv1.weight"] = np.zeros((64, 3, 5, 5), np.float32)          # too few elements for a 7x7: must be refused, not read
rc, _, _, err = pack(bad, "conv1", "", 64, 3, 4, 7)
assert rc == -1 and "conv1.weight" in err and "expected 9408" in err, err
bad = dict(rn)
bad["layer1.0.bn1.running_var"] = np.ones((3,), np.float32)         # truncated BatchNorm vector
rc, _, _, err = pack(bad, "layer1.0.conv1", "layer1.0.bn1", 64, 64, 64, 1)
assert rc == -1 and "running_var" in err, err
bad = dict(rn)
del bad["layer1.0.bn1.bias"]
rc, _, _, err = pack(bad, "layer1.0.conv1", "layer1.0.bn1", 64, 64, 64, 1)
assert rc == -1 and "missing key 'layer1.0.bn1.bias'" in err, err
# a NULL tensor pointer under a valid name
ptrs, names, numels, n, keep = marshal({"conv1.weight": rn["conv1.weight"]})
ptrs[0] = None
err = C.create_string_buffer(256)
out = np.zeros((64, 224), np.float32)
rc = lib.relax_host_pack_conv(ptrs, names, numels, n, b"conv1", b"", 64, 3, 4, 7, out.ctypes.data_as(C.c_void_p), None, err, 256)
assert rc == -1 and b"NULL" in err.value, err.value
# a short error buffer is respected
err = C.create_string_buffer(8)
rc = lib.relax_host_pack_conv(ptrs, names, numels, n, b"nope", b"", 64, 3, 4, 7, out.ctypes.data_as(C.c_void_p), None, err, 8)
assert rc == -1 and len(err.value) <= 7
print("malformed state dicts rejected with messages")

# quality head: fc1 + BatchNorm1d fold, 'module.' prefix stripped, 'n_averaged' ignored (src/demo_test.py:25-35)
F, Fpad = 35203, 35232
hd = synth.mlp_head_state_dict(F)
pref = {"module." + k: v for k, v in hd.items()}
pref["n_averaged"] = np.zeros((1,), np.float32)
ptrs, names, numels, n, keep = marshal(pref)
h1 = C.c_int()
err = C.create_string_buffer(256)
assert lib.relax_host_fold_fc_bn(ptrs, names, numels, n, F, Fpad, None, None, C.byref(h1), err, 256) == 0 and h1.value == 256
w1p = np.full((256, Fpad), np.nan, np.float32)
b1p = np.full((256,), np.nan, np.float32)
assert lib.relax_host_fold_fc_bn(ptrs, names, numels, n, F, Fpad, w1p.ctypes.data_as(C.c_void_p), b1p.ctypes.data_as(C.c_void_p),
                                 C.byref(h1), err, 256) == 0, err.value
s = (hd["bn1.weight"] / np.sqrt(hd["bn1.running_var"] + np.float32(1e-5))).astype(np.float32)
assert np.array_equal(w1p[:, :F], hd["fc1.weight"] * s[:, None]) and not w1p[:, F:].any()
assert np.array_equal(b1p, ((hd["fc1.bias"] - hd["bn1.running_mean"]) * s + hd["bn1.bias"]).astype(np.float32))
short = dict(pref)
short["module.bn1.running_mean"] = np.zeros((7,), np.float32)
ptrs, names, numels, n, keep = marshal(short)
assert lib.relax_host_fold_fc_bn(ptrs, names, numels, n, F, Fpad, w1p.ctypes.data_as(C.c_void_p), b1p.ctypes.data_as(C.c_void_p),
                                 C.byref(h1), err, 256) == -1 and b"bn1.running_mean" in err.value
assert lib.relax_host_fold_fc_bn(ptrs, names, numels, n, 1000, 1024, None, None, C.byref(h1), err, 256) == -1   # F does not divide
print("quality-head fold ok")

# tail split-K cost model: invariants over a sweep (what the launchers of gemm.hip / gemm_x6.hip rely on)
ft, ns = C.c_int(), C.c_int()
for slots in (256, 512, 768):
    for ntiles in list(range(1, 1400, 7)) + [256, 512, 594, 1024]:
        for nk in (2, 8, 48, 192, 288):
            for min_steps in (4, 8):
                for can in (0, 1):
                    lib.relax_host_tail_split(ntiles, slots, nk, min_steps, can, C.byref(ft), C.byref(ns))
                    rem = ntiles % slots
                    assert 1 <= ns.value <= 16
                    if ns.value == 1:
                        assert ft.value == ntiles
                    else:
                        assert can and rem > 0 and ft.value == ntiles - rem and nk // ns.value >= min_steps
                        assert -(-rem * ns.value // slots) / ns.value + 0.04 * ns.value < 1.0 - 0.05   # it had to pay
lib.relax_host_tail_split(594, 512, 48, 4, 1, C.byref(ft), C.byref(ns))        # ViT N = 768 on the fp32 kernel: 82 tail tiles
assert (ft.value, ns.value) == (512, 5), (ft.value, ns.value)   # 1/5 + 0.20 = 0.40 rounds (S = 6: 0.407)
lib.relax_host_tail_split(0, 256, 48, 8, 1, C.byref(ft), C.byref(ns))
assert (ft.value, ns.value) == (0, 1)
print("tail split-K model ok")
print("HOST_LOGIC_SANITIZED_OK")


# ---- the f16x2 scale helpers: powers of two, and bounds that BOUND (LayerNorm rows of any shape, Linear behind LayerNorm) ------------------
lib.relax_host_h2_scale_for_bound.restype = C.c_float
lib.relax_host_h2_scale_for_bound.argtypes = [C.c_double]
lib.relax_host_layernorm_out_bound.restype = C.c_double
lib.relax_host_linear_of_layernorm_bound.restype = C.c_double
for amax in (1e-30, 3e-5, 0.7, 1.0, 27.7, 65504.0, 1e30):
    sc = lib.relax_host_h2_scale_for_bound(amax)
    assert sc > 0 and np.log2(sc) == np.round(np.log2(sc)) and 2.0 ** 14 <= amax * sc < 2.0 ** 15 or amax in (1e-30, 1e30), (amax, sc)
for bad in (0.0, -1.0, float("inf"), float("nan")):
    assert lib.relax_host_h2_scale_for_bound(bad) == 1.0
rng = np.random.default_rng(3)
dim, nout = 192, 40
W = (rng.standard_normal((nout, dim)) * 0.3).astype(np.float32)
W[5] = 0
bias = rng.standard_normal(nout).astype(np.float32)
gamma = (rng.standard_normal(dim) * 2).astype(np.float32)
beta = rng.standard_normal(dim).astype(np.float32)
scales = np.zeros(nout, np.float32)
lib.relax_host_h2_weight_row_scales(W.ctypes.data_as(C.c_void_p), nout, dim, scales.ctypes.data_as(C.c_void_p))
rowmax = np.abs(W).max(axis=1)
assert scales[5] == 1.0 and all(2.0 ** 14 <= rowmax[n] * scales[n] < 2.0 ** 15 for n in range(nout) if n != 5)
ln_bound = lib.relax_host_layernorm_out_bound(gamma.ctypes.data_as(C.c_void_p), beta.ctypes.data_as(C.c_void_p), dim)
lin_bound = lib.relax_host_linear_of_layernorm_bound(W.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), gamma.ctypes.data_as(C.c_void_p),
                                                     beta.ctypes.data_as(C.c_void_p), dim, 8, nout)
worst_ln = worst_lin = 0.0
top = int(np.argmax(np.abs(gamma) * np.sqrt(dim - 1) + np.abs(beta)))      # the channel whose one-hot row attains the LayerNorm bound
rows = [rng.standard_normal(dim) * s + o for s, o in ((1, 0), (1e-3, 50), (100, -7))] + [np.eye(dim)[top] * 1e4, np.eye(dim)[top] * -1e4,
                                                                                           np.eye(dim)[3] * -5 + 1e-4 * rng.standard_normal(dim)]
for x in rows:
    z = (x - x.mean()) / np.sqrt(x.var() + 1e-6)
    y = z * gamma.astype(np.float64) + beta.astype(np.float64)
    worst_ln = max(worst_ln, np.abs(y).max())
    worst_lin = max(worst_lin, np.abs(W[8:].astype(np.float64) @ y + bias[8:]).max())
assert worst_ln <= ln_bound and worst_lin <= lin_bound, (worst_ln, ln_bound, worst_lin, lin_bound)
assert worst_ln > 0.5 * ln_bound          # (the one-hot row nearly attains the LayerNorm bound: it is tight, not merely safe)
print("f16x2 bound helpers ok: LayerNorm", worst_ln, "<=", ln_bound, " Linear", worst_lin, "<=", lin_bound)
