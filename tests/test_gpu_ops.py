"""Operator-level parity on the GPU: each HIP kernel against a plain fp32/fp64 torch CPU reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from relax_vqa_amd.engine import pack_conv_weight
from tests.gpu_common import assert_close, engine

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_kernel():
    """This file exercises the exact-fp32 contraction kernel (gemm.hip) and its tile variants; the default bf16x6 kernel has
    its operator-level tests in tests/test_gpu_x6.py."""
    engine().set_precision("fp32")
    yield


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 256, 512), (197 * 3, 2304, 768), (50, 64, 64),
                                   (4096, 64, 576), (777, 3072, 768), (12608, 768, 3072)])
def test_gemm_plain(M, N, K):
    A, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    want = (A.double() @ W.double().T).float().numpy()
    got = engine().op_gemm(A.cuda(), W.cuda())
    assert_close(got, want, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("variant", [1, 4, 5, 7, 10])
def test_gemm_tile_variants(variant):
    """Every selectable tile shape of the contraction kernel gives the same answer (N = 192 also exercises the N % 128 != 0 route)."""
    eng = engine()
    A, W, b = _rand(700, 256, seed=21), _rand(192, 256, seed=22, scale=1 / 16), _rand(192, seed=23)
    A2, W2 = _rand(1500, 128, seed=24), _rand(256, 128, seed=25, scale=0.09)
    want = torch.relu(A.double() @ W.double().T + b.double()).float().numpy()
    want2 = (A2.double() @ W2.double().T).float().numpy()
    eng.set_option("gemm_variant", variant)
    eng.set_option("gemm_variant_n64", variant if variant in (5, 10) else -1)
    try:
        assert_close(eng.op_gemm(A.cuda(), W.cuda(), b.cuda(), act=1), want, f"variant {variant} N=192")
        assert_close(eng.op_gemm(A2.cuda(), W2.cuda()), want2, f"variant {variant} N=256")
    finally:
        eng.set_option("gemm_variant", -1)
        eng.set_option("gemm_variant_n64", -1)


def test_gemm_identity_asymmetric():
    # A = I with an asymmetric W catches a transposed C write
    K = N = 64
    A = torch.eye(K)
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 100.0
    got = engine().op_gemm(A.cuda(), W.cuda()).cpu()
    assert torch.equal(got, W.T.contiguous())


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_epilogue(act):
    M, N, K = 333, 192, 96
    A, W, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=0.1), _rand(N, seed=5), _rand(M, N, seed=6)
    y = A.double() @ W.double().T + b.double() + r.double()
    want = [y, F.relu(y), F.gelu(y)][act].float().numpy()
    got = engine().op_gemm(A.cuda(), W.cuda(), b.cuda(), r.cuda(), act=act)
    assert_close(got, want, f"gemm epilogue act={act}")
    # in-place residual (out aliases residual), as the ViT blocks use it
    rr = r.cuda().clone()
    engine().op_gemm(A.cuda(), W.cuda(), b.cuda(), rr, act=act, out=rr)
    assert_close(rr, want, f"gemm in-place residual act={act}")


CONVS = [  # Nimg, H, Cin, Cout, k, stride, pad
    (2, 56, 64, 64, 1, 1, 0), (2, 56, 64, 64, 3, 1, 1), (2, 56, 128, 128, 3, 2, 1), (3, 28, 256, 512, 1, 2, 0),
    (2, 14, 256, 256, 3, 1, 1), (5, 7, 512, 512, 3, 1, 1), (2, 7, 2048, 512, 1, 1, 0), (1, 224, 4, 64, 7, 2, 3),
    (3, 30, 32, 64, 3, 2, 1),
]


@pytest.mark.parametrize("Nimg,H,Cin,Cout,k,stride,pad", CONVS)
def test_conv2d_nhwc(Nimg, H, Cin, Cout, k, stride, pad):
    x = _rand(Nimg, Cin, H, H, seed=7)
    w = _rand(Cout, Cin, k, k, seed=8, scale=(Cin * k * k) ** -0.5)
    b = _rand(Cout, seed=9)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)).float()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = torch.from_numpy(pack_conv_weight(w.numpy())).cuda()
    got = engine().op_conv2d_nhwc(x_nhwc, wp, b.cuda(), None, Cout, k, k, stride, pad, act=1)
    assert_close(got.permute(0, 3, 1, 2), want.numpy(), f"conv {Nimg}x{H}x{Cin}->{Cout} k{k}s{stride}")


def test_conv_residual_epilogue():
    x, w = _rand(2, 64, 14, 14, seed=10), _rand(256, 64, 1, 1, seed=11, scale=0.1)
    b, r = _rand(256, seed=12), _rand(2, 256, 14, 14, seed=13)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double()) + r.double()).float()
    got = engine().op_conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), torch.from_numpy(pack_conv_weight(w.numpy())).cuda(),
                                  b.cuda(), r.permute(0, 2, 3, 1).contiguous().cuda(), 256, 1, 1, 1, 0, act=1)
    assert_close(got.permute(0, 3, 1, 2), want.numpy(), "conv + residual + relu")


@pytest.mark.parametrize("rows,dim", [(197, 768), (1000, 192), (5, 384), (12608, 768)])
def test_layernorm(rows, dim):
    x, g, b = _rand(rows, dim, seed=14, scale=3.0) + 0.5, _rand(dim, seed=15) + 1.0, _rand(dim, seed=16)
    want = F.layer_norm(x.double(), (dim,), g.double(), b.double(), 1e-6).float().numpy()
    got = engine().op_layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6)
    assert_close(got, want, f"layernorm {rows}x{dim}")


@pytest.mark.parametrize("n_img,heads,scale", [(1, 3, 1.0), (3, 12, 1.0), (2, 6, 4.0)])
def test_attention(n_img, heads, scale):
    dim = heads * 64
    qkv = _rand(n_img * 197, 3 * dim, seed=17, scale=scale)
    t = qkv.double().reshape(n_img, 197, 3, heads, 64).permute(2, 0, 3, 1, 4)
    attn = ((t[0] @ t[1].transpose(-2, -1)) * 64 ** -0.5).softmax(dim=-1)
    want = (attn @ t[2]).transpose(1, 2).reshape(n_img * 197, dim).float().numpy()
    got = engine().op_attention(qkv.cuda(), n_img, heads)
    assert_close(got, want, f"attention n={n_img} heads={heads}")


def test_bn_relu_maxpool():
    x, sc, sh = _rand(3, 64, 112, 112, seed=18), _rand(64, seed=19), _rand(64, seed=20)
    y = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    want = F.max_pool2d(y, 3, 2, 1).numpy()
    got = engine().op_bn_relu_maxpool(x.permute(0, 2, 3, 1).contiguous().cuda(), sc.cuda(), sh.cuda())
    assert_close(got.permute(0, 3, 1, 2), want, "bn+relu+maxpool")


@pytest.mark.parametrize("n,hw,c", [(2, 12544, 64), (3, 3136, 256), (5, 49, 2048), (1, 196, 1024), (70, 784, 512)])
def test_gap(n, hw, c):
    x = _rand(n, hw, c, seed=21) + 0.3
    want = x.double().mean(dim=1).float().numpy()
    got = engine().op_gap(x.cuda())
    assert_close(got, want, f"gap {n}x{hw}x{c}")


# ---- opt-in bf16x3 precision (split products on the bf16 MFMA) -------------------------------------------------------
@pytest.fixture
def bf16x3():
    engine().set_precision("bf16x3")
    yield


@pytest.mark.parametrize("M,N,K", [(1000, 256, 512), (197 * 3, 2304, 768), (12608, 768, 3072), (50, 64, 64), (4096, 64, 576)])
def test_gemm_bf16x3_accuracy(bf16x3, M, N, K):
    A, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    want = (A.double() @ W.double().T).float().numpy()
    got = engine().op_gemm(A.cuda(), W.cuda()).cpu().numpy()
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 2e-5, rel                       # ~2^-16 per product, random signs
    assert_close(got, want, f"bf16x3 gemm {M}x{N}x{K}", rtol=1e-3, atol_frac=2e-4)


@pytest.mark.parametrize("Nimg,H,Cin,Cout,k,stride,pad", CONVS[:8])
def test_conv_bf16x3_accuracy(bf16x3, Nimg, H, Cin, Cout, k, stride, pad):
    x = _rand(Nimg, Cin, H, H, seed=7)
    w = _rand(Cout, Cin, k, k, seed=8, scale=(Cin * k * k) ** -0.5)
    b = _rand(Cout, seed=9)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)).float().numpy()
    got = engine().op_conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), torch.from_numpy(pack_conv_weight(w.numpy())).cuda(),
                                  b.cuda(), None, Cout, k, k, stride, pad, act=1).permute(0, 3, 1, 2).cpu().numpy()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-5


def test_gemm_bf16x3_large_tile_path(bf16x3):
    """Enough 256x256 tiles (>= 256) to take the 8-wave large-tile kernel: ragged M, tail split-K, fused epilogue."""
    M, N, K = 25001, 768, 768
    A, W, b, r = _rand(M, K, seed=11), _rand(N, K, seed=12, scale=K ** -0.5), _rand(N, seed=13), _rand(M, N, seed=14)
    want = F.gelu((A.double().cuda() @ W.double().cuda().T) + b.double().cuda() + r.double().cuda()).float().cpu().numpy()
    got = engine().op_gemm(A.cuda(), W.cuda(), b.cuda(), r.cuda(), act=2).cpu().numpy()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-5
    assert_close(got, want, "bf16x3 large-tile gemm", rtol=1e-3, atol_frac=2e-4)
    fp32_rows = engine().op_gemm(A[:300].cuda(), W.cuda(), b.cuda(), r[:300].cuda(), act=2).cpu().numpy()   # small-tile path
    assert np.linalg.norm(fp32_rows - got[:300]) / np.linalg.norm(got[:300]) < 2e-5
