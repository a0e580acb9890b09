"""f16x2 (csrc/gemm_h2.hip, csrc/h2.h): fp32 operands as two fp16 planes of a power-of-two multiple of themselves, all four partial
products in two v_mfma_f32_16x16x32_f16, fp32 accumulate.  The claim under test is fp32-GRADE accuracy: against an fp64 reference the
error is no larger than the exact-fp32 path's (an fp32 FMA chain) - the gate of the round-4 review - and the scales can never
overflow (they come from bounds, not from data)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fragment_ref, pooling_ref, vit_ref
from tests.gpu_common import WEIGHT_SET_IDS, WEIGHT_SETS, assert_close, engine, golden_tag, synth, vit_weights

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.fixture()
def h2():
    eng = engine()
    eng.set_precision("f16x2")
    assert eng.precision() == "f16x2"
    yield eng            # (the autouse fixture restores the engine's precision)


def _two_plane_value(A):
    """numpy emulation of csrc/h2.h per ROW of A: scale = the power of two that puts the row maximum into [2^14, 2^15), hi = fp16(x s),
    lo = fp16(x s - hi); returns (hi + lo) / s in double - what the planes hold."""
    A = np.asarray(A, dtype=np.float32)
    amax = np.abs(A).max(axis=1, keepdims=True)
    _, e = np.frexp(amax)
    s = np.where(amax > 0, np.exp2((15 - e).astype(np.float64)), 1.0).astype(np.float32)
    v = (A * s).astype(np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return (hi.astype(np.float64) + lo.astype(np.float64)) / s.astype(np.float64)


def test_permutation_matrix_copies_the_two_plane_value_bit_for_bit(h2):
    """W = a permutation matrix (its rows become 2^14 exactly): out[m, n] = the two-plane value of A[m, perm[n]] EXACTLY - hi x 2^14
    + lo x 2^14 is exact in the fp32 accumulator and the scales are powers of two.  One equality checks the plane layout, the LDS
    image, the DMA piece map, the per-row scales, fp16 subnormal lo planes and the C write; the asymmetric permutation catches any
    transpose, M = 300 the zero-filled rows past M, the 2^-6 .. 2^6 spread inside a row the part of the window below 2^-3."""
    M, K = 300, 512
    A = (_rand(M, K, seed=1) * torch.logspace(-6, 6, K, base=2.0)[None, :]).float()
    want = _two_plane_value(A.numpy())
    assert np.abs(want - A.double().numpy()).max() > 0          # (the format does round: the test is not vacuous)
    for N in (256, 512):
        perm = torch.randperm(K, generator=torch.Generator().manual_seed(N))[:N]
        W = torch.zeros(N, K)
        W[torch.arange(N), perm] = 1.0
        got = h2.op_gemm(A.cuda(), W.cuda()).cpu().double().numpy()
        assert np.array_equal(got, want[:, perm.numpy()]), f"N={N}"
    # and the relative error of what the planes hold: 2^-22 of each value that sits in the window of its row
    big = np.abs(A.numpy()) * 2.0 ** 17 >= np.abs(A.numpy()).max(axis=1, keepdims=True)
    rel = np.abs(want - A.double().numpy())[big] / np.abs(A.double().numpy())[big]
    assert rel.max() <= 2.0 ** -22


@pytest.mark.parametrize("M,N,K", [(128, 256, 32), (1000, 256, 512), (197 * 3, 2304, 768), (777, 3072, 768), (12608, 768, 3072),
                                   (2049, 512, 4608), (256 * 40, 768, 768), (700, 256, 256), (300, 256, 64)])
@pytest.mark.parametrize("form", [1, 0, 2], ids=["k32_three_products", "k16_four_products", "k32_four_products"])
def test_error_is_no_larger_than_the_fp32_paths(M, N, K, form):
    eng = engine()
    eng.set_option("h2_form", form)
    A, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    ref = A.double() @ W.double().T
    eng.set_precision("fp32")
    e32 = (eng.op_gemm(A.cuda(), W.cuda()).cpu().double() - ref).abs()
    eng.set_precision("bf16x6")
    e6 = (eng.op_gemm(A.cuda(), W.cuda()).cpu().double() - ref).abs()
    eng.set_precision("f16x2")
    got = eng.op_gemm(A.cuda(), W.cuda())
    again = eng.op_gemm(A.cuda(), W.cuda())
    assert torch.equal(got, again)
    e2 = (got.cpu().double() - ref).abs()
    scale = ref.abs().mean().item()
    print(f"\n{M}x{N}x{K}: mean |err| / mean |ref|: fp32 {e32.mean().item() / scale:.3e}  bf16x6 {e6.mean().item() / scale:.3e}  "
          f"f16x2 {e2.mean().item() / scale:.3e};  max: fp32 {e32.max().item() / scale:.3e}  bf16x6 {e6.max().item() / scale:.3e}  "
          f"f16x2 {e2.max().item() / scale:.3e}")
    assert_close(got, ref.float().numpy(), f"f16x2 gemm {M}x{N}x{K}")
    eng.set_option("h2_form", 1)
    assert e2.mean().item() <= 1.05 * e32.mean().item() + 1e-12, "f16x2 mean error exceeds the fp32 FMA chain's"
    assert e2.max().item() <= 1.5 * e32.max().item() + 1e-12, "f16x2 worst error exceeds the fp32 FMA chain's"


def test_rows_of_very_different_size_and_sparse_rows(h2):
    """Per-row scales (operator level): rows whose magnitudes differ by 2^60, an all-zero row, a row with a single nonzero, and values
    spread over 2^-12 .. 2^0 inside a row.  Error measured against the sum of magnitudes, beside the fp32 FMA chain."""
    eng = h2
    M, N, K = 300, 256, 768
    g = torch.Generator().manual_seed(5)
    A = _rand(M, K, seed=3) * torch.exp2(torch.randint(-12, 1, (M, K), generator=g).float()) * torch.exp2(torch.randint(-30, 31, (M, 1), generator=g).float())
    A[7] = 0
    A[9] = 0
    A[9, 100] = 3.0e-20
    W = _rand(N, K, seed=4) * torch.exp2(torch.randint(-8, 1, (N, K), generator=g).float())
    ref = A.double() @ W.double().T
    mag = (A.double().abs() @ W.double().abs().T).clamp_min(1e-300)
    got = eng.op_gemm(A.cuda(), W.cuda()).cpu().double()
    eng.set_precision("fp32")
    e32 = ((eng.op_gemm(A.cuda(), W.cuda()).cpu().double() - ref).abs() / mag)
    e2 = ((got - ref).abs() / mag)
    print(f"\nerr / sum|a||w|: fp32 mean {e32.mean().item():.3e} max {e32.max().item():.3e}; f16x2 mean {e2.mean().item():.3e} max {e2.max().item():.3e}")
    assert torch.isfinite(got).all() and torch.all(got[7] == 0)
    assert e2.max().item() < 1e-6 and e2.mean().item() <= 1.05 * e32.mean().item() + 1e-9


@pytest.mark.parametrize("act", [0, 1, 2])
def test_epilogue(h2, act):
    M, N, K = 333, 256, 96
    A, W, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=0.1), _rand(N, seed=5), _rand(M, N, seed=6)
    y = A.double() @ W.double().T + b.double() + r.double()
    want = [y, F.relu(y), F.gelu(y)][act].float().numpy()
    assert_close(h2.op_gemm(A.cuda(), W.cuda(), b.cuda(), r.cuda(), act=act), want, f"f16x2 epilogue act={act}")
    rr = r.cuda().clone()
    h2.op_gemm(A.cuda(), W.cuda(), b.cuda(), rr, act=act, out=rr)
    assert_close(rr, want, f"f16x2 in-place residual act={act}")


def test_shapes_the_tile_does_not_take_run_bf16x6(h2):
    """N % 256 != 0 (the 64 / 128-column tiles) stays on the split-plane kernel under "gemm_precision" 3: same bits as under 2."""
    A, W = _rand(500, 256, seed=1).cuda(), _rand(192, 256, seed=2, scale=0.06).cuda()
    got = h2.op_gemm(A, W)
    h2.set_precision("bf16x6")
    assert torch.equal(got, h2.op_gemm(A, W))


def test_split_k_is_deterministic_and_optional_and_stages_agree(h2):
    """300 tiles of 256x256: 44 tail tiles are cut along K.  Same bits run to run; with gemm_split_k = 0 the bits do not depend on how
    many rows travel together; the 3- and 4-stage forms of the loop give the same bits (same products, same order)."""
    M, N, K = 256 * 100, 768, 768
    A, W = _rand(M, K, seed=8).cuda(), _rand(N, K, seed=9, scale=K ** -0.5).cuda()
    a = h2.op_gemm(A, W)
    assert torch.equal(a, h2.op_gemm(A, W))
    h2.set_option("h2_form", 0)          # the 16-k form of the loop: 3 or 4 LDS stages
    try:
        b3 = h2.op_gemm(A, W)
        h2.set_option("h2_stages", 4)
        assert torch.equal(b3, h2.op_gemm(A, W))
    finally:
        h2.set_option("h2_stages", 3)
        h2.set_option("h2_form", 1)
    assert_close(b3, a.cpu().numpy(), "16-k four-product form vs 32-k three-product form", rtol=1e-4, atol_frac=1e-5)
    h2.set_option("gemm_split_k", 0)
    try:
        whole = h2.op_gemm(A, W)
        part = h2.op_gemm(A[: 256 * 7], W)
        odd = h2.op_gemm(A[5: 5 + 777].contiguous(), W)
    finally:
        h2.set_option("gemm_split_k", 1)
    assert torch.equal(whole[: 256 * 7], part) and torch.equal(whole[5: 5 + 777], odd)
    assert_close(a, whole.cpu().numpy(), "split-K vs whole-K", rtol=1e-4, atol_frac=1e-5)


# ---- convolutions under f16x2 (the geometries of ResNet-50 layer3 / layer4: Cin % 32 == 0, Cout % 256 == 0) -----------------------------
from relax_vqa_amd.engine import pack_conv_weight  # noqa: E402

CONVS = [  # Nimg, H, Cin, Cout, k, stride, pad
    (2, 14, 256, 256, 3, 1, 1), (5, 7, 512, 512, 3, 1, 1), (2, 7, 2048, 512, 1, 1, 0), (3, 28, 256, 512, 1, 2, 0), (7, 7, 512, 2048, 1, 1, 0),
    (2, 14, 1024, 256, 1, 1, 0), (3, 14, 512, 512, 3, 2, 1), (40, 14, 256, 256, 3, 1, 1), (3, 30, 32, 256, 3, 2, 1),
    # small images: a 256-row tile spans 4 / 11 / 16 images of very different magnitude (the epilogue finds a row's image without a division)
    (33, 9, 64, 256, 1, 1, 0), (50, 5, 64, 256, 3, 1, 1), (70, 4, 64, 256, 1, 1, 0),
    # the four-wave f16x2 form of gemm_x6.hip (64 / 128 output columns, K x K, K >= 256): the 3x3 convolutions of layer1 / layer2
    (2, 56, 64, 64, 3, 1, 1), (3, 28, 128, 128, 3, 1, 1), (2, 56, 128, 128, 3, 2, 1), (5, 9, 64, 64, 3, 1, 1), (2, 14, 32, 64, 3, 1, 1),
    (37, 7, 64, 128, 3, 1, 1), (2, 12, 16, 64, 5, 2, 2),
]


@pytest.mark.parametrize("Nimg,H,Cin,Cout,k,stride,pad", CONVS)
def test_conv2d_nhwc_under_f16x2(Nimg, H, Cin, Cout, k, stride, pad):
    """gemm_h3 as implicit GEMM (taps, strides, padding by the DMA's range check, rows of a tile spanning several images) with one scale
    per IMAGE: images of very different magnitude in one batch (x 2^-12 .. 2^12) keep their own precision.  Error against fp64 no
    larger than the exact-fp32 path's."""
    eng = engine()
    x = _rand(Nimg, Cin, H, H, seed=7) * torch.exp2(torch.linspace(-12, 12, Nimg))[:, None, None, None]
    w = _rand(Cout, Cin, k, k, seed=8, scale=(Cin * k * k) ** -0.5)
    b = _rand(Cout, seed=9) * 1e-3
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = torch.from_numpy(pack_conv_weight(w.numpy())).cuda()
    eng.set_precision("fp32")
    e32 = (eng.op_conv2d_nhwc(x_nhwc, wp, b.cuda(), None, Cout, k, k, stride, pad, act=1).permute(0, 3, 1, 2).cpu().double() - ref).abs()
    eng.set_precision("f16x2")
    got = eng.op_conv2d_nhwc(x_nhwc, wp, b.cuda(), None, Cout, k, k, stride, pad, act=1).permute(0, 3, 1, 2)
    assert torch.equal(got, eng.op_conv2d_nhwc(x_nhwc, wp, b.cuda(), None, Cout, k, k, stride, pad, act=1).permute(0, 3, 1, 2))
    e2 = (got.cpu().double() - ref).abs()
    per_img = ref.abs().mean(dim=(1, 2, 3)).clamp_min(1e-300)
    r32, r2 = (e32.mean(dim=(1, 2, 3)) / per_img), (e2.mean(dim=(1, 2, 3)) / per_img)
    print(f"\nconv {Nimg}x{H}x{Cin}->{Cout} k{k}s{stride}: per-image mean err / mean |ref|: fp32 {r32.mean().item():.3e} (max {r32.max().item():.3e})  "
          f"f16x2 {r2.mean().item():.3e} (max {r2.max().item():.3e})")
    for i in range(Nimg):
        assert_close(got[i], ref[i].float().numpy(), f"f16x2 conv image {i}")
    assert r2.mean().item() <= 1.05 * r32.mean().item() + 1e-12 and r2.max().item() <= 1.25 * r32.max().item() + 1e-12


# ---- the ViT under f16x2 ------------------------------------------------------------------------------------------------
def _fragments(n, seed=0):
    frs = []
    for i in range(n):
        o, nx = synth.synthetic_pair(240, 320, 500 + seed * 64 + i)
        f = fragment_ref.fragment_pair(o, nx)
        frs.append(f["ori_frag"] if i % 2 == 0 else f["diff_frag"])
    return np.stack(frs)


@pytest.mark.parametrize("adv", WEIGHT_SETS, ids=WEIGHT_SET_IDS)
def test_vit_base_matches_reference_golden_tokens_under_f16x2(golden_dir, h2, adv):
    """The reference's own VisionTransformer outputs (tests/golden/vit_base{,_adv,_out}_tokens.npz): regular weights, the adversarial
    set (logits of +-20, LayerNorm gains of mixed sign) and the outlier set (five residual-stream channels hundreds of times the
    median with LayerNorm gains up to 10 on them: the static bounds are 2^7 - 2^10 loose for the other 763 channels)."""
    vit_weights("vit_base", adversarial=adv)
    z = np.load(os.path.join(golden_dir, f"vit_base{golden_tag(adv)}_tokens.npz"))
    tokens, pooled = h2.vit_features(torch.from_numpy(z["frags"]).cuda(), tokens=True, pooled=True)
    assert_close(tokens, z["tokens"], "vit_base tokens (f16x2) vs reference VisionTransformer")
    want = np.stack([pooling_ref.vit_pool_vector(t) for t in z["tokens"]])
    assert_close(pooled, want, "vit_base pooled (f16x2) vs reference process_video_feature")


@pytest.mark.parametrize("adv", WEIGHT_SETS, ids=WEIGHT_SET_IDS)
def test_vit_base_error_against_fp64_is_no_larger_than_the_fp32_paths(adv):
    """12 blocks deep: tokens of the exact-fp32 path, of bf16x6 and of f16x2 against an fp64 run of the oracle (same fp32 weights and
    inputs, all arithmetic in double).  The gate: f16x2 no further from fp64 than the fp32 FMA chain - on the outlier set also over
    the QUIET channels alone (|token| < 10: the 763 channels the five outlier channels would otherwise hide in a norm), which is
    where a loose static scale would show (values far below the scale's top keep an absolute, not a relative, error)."""
    sd = vit_weights("vit_base", adversarial=adv)
    eng = engine()
    frags = _fragments(3, seed=2)
    sd64 = {k: v.double() for k, v in vit_ref.to_torch_state_dict(sd).items()}
    ref = vit_ref.forward_tokens(sd64, vit_ref.preprocess_bgr_u8(frags).double(), 12).numpy()
    f = torch.from_numpy(frags).cuda()
    out = {}
    for prec in ("fp32", "bf16x6", "f16x2"):
        eng.set_precision(prec)
        t, _ = eng.vit_features(f, tokens=True, pooled=False)
        t2, _ = eng.vit_features(f, tokens=True, pooled=False)
        assert torch.equal(t, t2), f"{prec} is not deterministic"
        e = np.abs(t.cpu().numpy().astype(np.float64) - ref)
        quiet = np.abs(ref).max(axis=(0, 1)) < 10.0
        out[prec] = (np.linalg.norm(e) / np.linalg.norm(ref), e.max(), np.linalg.norm(e[..., quiet]) / np.linalg.norm(ref[..., quiet]))
    print(f"\nvit_base tokens ({WEIGHT_SET_IDS[WEIGHT_SETS.index(adv)]} weights) vs fp64 (norm-rel, max abs, norm-rel over the {int(quiet.sum())} quiet channels): "
          + "  ".join(f"{k} {v[0]:.3e} {v[1]:.3e} {v[2]:.3e}" for k, v in out.items()))
    assert out["f16x2"][0] <= 1.1 * out["fp32"][0] and out["f16x2"][1] <= 1.5 * out["fp32"][1]
    assert out["f16x2"][2] <= 1.1 * out["fp32"][2], "f16x2 loses to the fp32 chain on the channels outside the outliers"


def test_vit_rows_do_not_depend_on_the_batch_under_f16x2(h2):
    """Static scales: with the tail split off an image's tokens are the same bits alone, first or last in a batch of 37."""
    vit_weights("vit_base")
    f = torch.from_numpy(_fragments(5, seed=3)).cuda()
    big = f.repeat(8, 1, 1, 1)[:37]
    h2.set_option("gemm_split_k", 0)
    try:
        _, alone = h2.vit_features(f[2:3], tokens=False, pooled=True)
        _, five = h2.vit_features(f, tokens=False, pooled=True)
        _, many = h2.vit_features(big, tokens=False, pooled=True)
    finally:
        h2.set_option("gemm_split_k", 1)
    assert torch.equal(alone[0], five[2]) and torch.equal(five[2], many[2]) and torch.equal(many[2], many[32])


def test_extreme_inputs_cannot_overflow_the_static_scales():
    """All-white, all-black and a checkerboard of 0 / 255 (the largest patch-embedding outputs and the most uneven LayerNorm rows a
    uint8 input can make), regular and adversarial weights: the tokens stay finite, and against an fp64 run of the oracle f16x2 is no
    further away than the exact-fp32 path (the adversarial network on a checkerboard is ill-conditioned - its one-hot softmax rows
    amplify any fp32 rounding to 1e-4 - which is why the comparison is against fp64 and beside the FMA chain, not against a bar)."""
    eng = engine()
    for adv in (False, True):
        sd = vit_weights("vit_base", adversarial=adv)
        frags = np.zeros((3, 224, 224, 3), dtype=np.uint8)
        frags[0] = 255
        yy, xx = np.mgrid[0:224, 0:224]
        frags[2] = (((yy // 16 + xx // 16) % 2) * 255)[:, :, None]
        sd64 = {k: v.double() for k, v in vit_ref.to_torch_state_dict(sd).items()}
        ref = vit_ref.forward_tokens(sd64, vit_ref.preprocess_bgr_u8(frags).double(), 12).numpy()
        f = torch.from_numpy(frags).cuda()
        err = {}
        for prec in ("fp32", "f16x2"):
            eng.set_precision(prec)
            tokens, _ = eng.vit_features(f, tokens=True, pooled=False)
            assert torch.isfinite(tokens).all(), prec
            e = np.abs(tokens.cpu().numpy().astype(np.float64) - ref)
            err[prec] = (np.linalg.norm(e) / np.linalg.norm(ref), e.max())
        print(f"\nextreme inputs, adversarial={adv}: vs fp64 (norm-rel, max abs): fp32 {err['fp32'][0]:.3e} {err['fp32'][1]:.3e}  "
              f"f16x2 {err['f16x2'][0]:.3e} {err['f16x2'][1]:.3e}")
        assert err["f16x2"][0] <= 1.25 * err["fp32"][0] and err["f16x2"][1] <= 2.0 * err["fp32"][1]
        if not adv:
            assert_close(tokens, ref.astype(np.float32), "extreme inputs, regular weights")


@pytest.mark.parametrize("n_img,heads,scale", [(1, 3, 1.0), (3, 12, 1.0), (2, 6, 4.0), (40, 12, 2.0), (5, 12, 0.02)])
def test_attention_under_f16x2(n_img, heads, scale):
    """softmax(q k^T / 8) v on two fp16 planes with three partial products (csrc/attention_h2.hip: K and V by LDS-DMA straight into the
    fragment images, V read with the transposing LDS read) against fp64, beside the fp32-MFMA kernel and bf16x6: scale 4 makes logits of
    +-60 (near one-hot rows), 0.02 nearly uniform rows of tiny values; 40 x 12 = 480 items > 256 workgroups exercises the persistent loop.
    The gate is the review's: error no larger than the exact-fp32 path's."""
    eng = engine()
    dim = heads * 64
    qkv = _rand(n_img * 197, 3 * dim, seed=17, scale=scale)
    t = qkv.double().reshape(n_img, 197, 3, heads, 64).permute(2, 0, 3, 1, 4)
    attn = ((t[0] @ t[1].transpose(-2, -1)) * 64 ** -0.5).softmax(dim=-1)
    ref = (attn @ t[2]).transpose(1, 2).reshape(n_img * 197, dim)
    eng.set_precision("fp32")
    e32 = (eng.op_attention(qkv.cuda(), n_img, heads).cpu().double() - ref).abs()
    eng.set_precision("bf16x6")
    e6 = (eng.op_attention(qkv.cuda(), n_img, heads).cpu().double() - ref).abs()
    eng.set_precision("f16x2")
    assert eng.get_option("att_h2") == 1
    got = eng.op_attention(qkv.cuda(), n_img, heads)
    assert torch.equal(got, eng.op_attention(qkv.cuda(), n_img, heads))
    assert_close(got, ref.float().numpy(), f"f16x2 attention n={n_img} heads={heads}")
    e2 = (got.cpu().double() - ref).abs()
    print(f"\nattention {n_img}x{heads} scale {scale}: mean err fp32 {e32.mean().item():.3e} bf16x6 {e6.mean().item():.3e} f16x2 {e2.mean().item():.3e}; "
          f"max fp32 {e32.max().item():.3e} bf16x6 {e6.max().item():.3e} f16x2 {e2.max().item():.3e}")
    assert e2.mean().item() <= 1.05 * e32.mean().item() + 1e-12 and e2.max().item() <= 1.5 * e32.max().item() + 1e-12
    try:
        eng.set_option("att_h2", 0)    # the A/B switch: attention_x6 under gemm_precision 3
        old = eng.op_attention(qkv.cuda(), n_img, heads)
    finally:
        eng.set_option("att_h2", 1)
    assert_close(old, ref.float().numpy(), "attention_x6 under f16x2 (att_h2 = 0)")


# ---- ResNet-50 under f16x2: layer3 / layer4 on fp16 planes with per-image scales ------------------------------------------------------------
from oracle import resnet50_ref  # noqa: E402
from tests.gpu_common import rn50_weights  # noqa: E402,F811


@pytest.mark.parametrize("adv", WEIGHT_SETS, ids=WEIGHT_SET_IDS)
def test_resnet50_every_tap_under_f16x2_and_error_against_fp64(adv):
    """All 15 taps + both feature vectors with layer3 / layer4 on the f16x2 kernels ("rn_h2", the default under gemm_precision 3): inside
    the bar against the fp32 oracle, deterministic, and against an fp64 run of the oracle no further away than bf16x6 everywhere (x 1.25) and
    than torch-CPU fp32 - the reference's own arithmetic - on every tap.  Images of very different brightness share the batch: the scales are
    per image.  The adversarial set (BatchNorm variances 1e-3 .. 10: activations of very different size from layer to layer) exercises
    the Hoelder bounds; the outlier set (one channel per stage 50 - 100 x the others, dead channel groups) makes the per-image maximum -
    and so the scale - the business of ONE channel: the gate is then also taken over the other channels alone."""
    sd = rn50_weights(adversarial=adv)
    eng = engine()
    frags = _fragments(4)
    frags[1] = frags[1] // 16                      # a dark image and a bright one in the same batch
    frags[2] = 255 - frags[2] // 8
    f = torch.from_numpy(frags).cuda()
    eng.set_precision("f16x2")
    assert eng.get_option("rn_h2") == 1 and eng.get_option("rn_h2_early") == 1
    ls2, pool2, taps2 = eng.resnet50_features(f, taps=range(15))
    ls2b, pool2b = eng.resnet50_features(f)
    assert torch.equal(ls2, ls2b) and torch.equal(pool2, pool2b), "f16x2 ResNet-50 is not deterministic"
    assert eng.get_option("rn_fuse") == 1
    try:
        eng.set_option("rn_fuse", 0)          # conv2 and conv3 of the layer1 / layer2 blocks as two launches (conv3 on bf16x6)
        _, _, taps2nofuse = eng.resnet50_features(f, taps=range(15))
        eng.set_option("rn_fuse", 1)
        eng.set_option("rn_c1_h2", 0)         # layer2's conv1 on bf16x6 (fp32 rows split into three bf16 planes) instead of two fp16 planes
        _, _, taps2c1x6 = eng.resnet50_features(f, taps=range(15))
        eng.set_option("rn_c1_h2", 1)
        eng.set_option("rn_h2_early", 0)      # layer3 / layer4 only
        _, _, taps2late = eng.resnet50_features(f, taps=range(15))
        eng.set_option("rn_h2", 0)
        ls6, pool6, taps6 = eng.resnet50_features(f, taps=range(15))
    finally:
        eng.set_option("rn_h2", 1)
        eng.set_option("rn_h2_early", 1)
        eng.set_option("rn_fuse", 1)
        eng.set_option("rn_c1_h2", 1)
    tsd = resnet50_ref.to_torch_state_dict(sd)
    ref_taps, _ = resnet50_ref.forward_taps(tsd, resnet50_ref.preprocess_bgr_u8(frags))
    sd64 = {k: v.double() for k, v in tsd.items()}
    ref64, _ = resnet50_ref.forward_taps(sd64, resnet50_ref.preprocess_bgr_u8(frags).double())

    def rel(a, r):
        return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - r) / np.linalg.norm(r))

    for i, name in enumerate(pooling_ref.RESNET50_TAPS):
        assert_close(taps2[i], ref_taps[name].numpy(), f"f16x2 {name}", channel_axis=1)
        r = ref64[name].numpy()
        n2, n6, ncpu = rel(taps2[i].cpu().numpy(), r), rel(taps6[i].cpu().numpy(), r), rel(ref_taps[name].numpy(), r)
        n2l = rel(taps2late[i].cpu().numpy(), r)
        n2u = rel(taps2nofuse[i].cpu().numpy(), r)
        print(f"{name:22s} vs fp64: f16x2 (fused blocks + 3x3s of layers 1-2 + layers 3-4) {n2:.3e}  the same unfused {n2u:.3e}  f16x2 layers 3-4 only {n2l:.3e}  "
              f"bf16x6 everywhere {n6:.3e}  torch CPU fp32 {ncpu:.3e}")
        assert n2 <= 1.25 * n6 + 1e-9 and n2 <= ncpu, name
        assert n2u <= 1.25 * n6 + 1e-9 and n2u <= ncpu, name
        n2c = rel(taps2c1x6[i].cpu().numpy(), r)      # (layer2's conv1 back on bf16x6: the same gates)
        assert n2c <= 1.25 * n6 + 1e-9 and n2c <= ncpu, f"{name}: {n2c:.3e}"
        assert n2l <= 1.25 * n6 + 1e-9 and n2l <= ncpu, name
        if adv == "outliers" and i > 0:     # the same gate without the hot channel (it carries most of a norm over all channels)
            cm = np.abs(r).max(axis=(0, 2, 3))
            q = cm < 0.25 * cm.max()
            q2, q6, qcpu = (rel(np.asarray(a)[:, q], r[:, q]) for a in (taps2[i].cpu().numpy(), taps6[i].cpu().numpy(), ref_taps[name].numpy()))
            print(f"{'':22s} without the hot channel(s) ({int(q.sum())} of {q.size}): f16x2 {q2:.3e}  bf16x6 {q6:.3e}  torch CPU fp32 {qcpu:.3e}")
            assert q2 <= 1.25 * q6 + 1e-9 and q2 <= qcpu, name + " (quiet channels)"
        if i < 7:   # (layer2[3], the hand-over block, runs unsplit with its fp16-plane output: same kernel, possibly another K-slice order)
            assert torch.equal(taps2late[i], taps6[i]), f"{name}: without rn_h2_early layer1 / layer2 run the same kernels either way"
    assert_close(ls2, resnet50_ref.layer_stack_features(tsd, frags), "f16x2 layer-stack")
    assert_close(pool2, resnet50_ref.pool_features(tsd, frags), "f16x2 pool")


def test_back_to_back_blocks_give_the_same_bits_on_both_tile_heights(h2):
    """"b2b_rows": the fused conv2 -> conv3 launches on 128-row tiles (three workgroups per CU: measured slower, 2.49 against 2.13 ms per launch) and on 256-row tiles (the default): every
    row's arithmetic is its own (one scale per pixel row, 16-row groups of the fused mean), so the features are the same bits."""
    rn50_weights()
    f = torch.from_numpy(_fragments(3, seed=5)).cuda()
    assert h2.get_option("rn_fuse") == 1 and h2.get_option("b2b_rows") == 256
    ls_a, pool_a, taps_a = h2.resnet50_features(f, taps=range(15))
    try:
        h2.set_option("b2b_rows", 128)
        ls_b, pool_b, taps_b = h2.resnet50_features(f, taps=range(15))
    finally:
        h2.set_option("b2b_rows", 256)
    assert torch.equal(ls_a, ls_b) and torch.equal(pool_a, pool_b)
    for i in range(15):
        assert torch.equal(taps_a[i], taps_b[i]), i


def test_resnet50_rows_do_not_depend_on_the_batch_under_f16x2(h2):
    """Per-image scales from per-image maxima (integer atomic max: order-free): with the tail split off an image's features are the same
    bits alone, among 5, and among 37 images."""
    rn50_weights()
    f = torch.from_numpy(_fragments(5, seed=4)).cuda()
    big = f.repeat(8, 1, 1, 1)[:37]
    h2.set_option("gemm_split_k", 0)
    try:
        ls1, p1 = h2.resnet50_features(f[2:3])
        ls5, p5 = h2.resnet50_features(f)
        ls37, p37 = h2.resnet50_features(big)
    finally:
        h2.set_option("gemm_split_k", 1)
    assert torch.equal(ls1[0], ls5[2]) and torch.equal(ls5[2], ls37[2]) and torch.equal(ls37[2], ls37[32])
    assert torch.equal(p1[0], p5[2]) and torch.equal(p5[2], p37[2]) and torch.equal(p37[2], p37[32])


def test_resnet50_extreme_images_under_f16x2():
    """All-black, all-white and checkerboard fragments (the per-image maxima at their extremes: a black image's layer3 input is whatever the
    biases leave, possibly zero everywhere): finite features inside the bar against the oracle, printed beside bf16x6."""
    sd = rn50_weights()
    eng = engine()
    frags = np.zeros((4, 224, 224, 3), dtype=np.uint8)
    frags[1] = 255
    yy, xx = np.mgrid[0:224, 0:224]
    frags[2] = (((yy // 16 + xx // 16) % 2) * 255)[:, :, None]
    frags[3] = _fragments(1)[0]
    f = torch.from_numpy(frags).cuda()
    tsd = resnet50_ref.to_torch_state_dict(sd)
    want_ls = resnet50_ref.layer_stack_features(tsd, frags)
    want_pool = resnet50_ref.pool_features(tsd, frags)
    eng.set_precision("f16x2")
    ls2, pool2 = eng.resnet50_features(f)
    assert torch.isfinite(ls2).all() and torch.isfinite(pool2).all()
    assert_close(ls2, want_ls, "extreme images, layer stack (f16x2)")
    assert_close(pool2, want_pool, "extreme images, pool (f16x2)")
    eng.set_precision("bf16x6")
    ls6, pool6 = eng.resnet50_features(f)
    d2 = np.abs(ls2.cpu().numpy().astype(np.float64) - want_ls).max(axis=1)
    d6 = np.abs(ls6.cpu().numpy().astype(np.float64) - want_ls).max(axis=1)
    print(f"\nextreme images: max |layer stack - oracle| per image: f16x2 {d2}  bf16x6 {d6}")
