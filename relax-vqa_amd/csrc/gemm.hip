// The contraction kernel of the backbones: fp32 implicit-GEMM convolution / GEMM on the gfx950
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + residual[m, n] )
//
//   m = output pixel (image, oy, ox) of an NHWC activation  (or a plain GEMM row)
//   k = (dy*KW + dx)*Cin + c, gathered on the fly from the NHWC input (never materialised)
//   n = output channel; W is stored [Cout][Kpad] (K fastest) so A and W tiles stage identically
//
// Tiling for 64-wide wavefronts: a workgroup of 4 waves owns a BM x BN tile, each wave a
// (TM*32) x (TN*32) sub-tile held in TM*TN 32x32 MFMA accumulators.  The K loop runs BK = 32
// deep steps through two LDS buffers (register-staged prefetch of step k+1 under the MFMAs of
// step k, one barrier per step).  LDS rows are padded to 36 floats so both the ds_write_b128
// staging writes and the ds_read_b128 fragment reads are bank-conflict free.  Because the MFMA
// sums over k, the k order inside a step is free: lane (i, h) reads the 4 consecutive floats
// k = 8q+4h .. 8q+4h+3 of row i with ONE 128-bit LDS read and feeds them to 4 MFMAs.
// Workgroup ids are remapped so that the workgroups sharing one XCD (and its L2) walk
// neighbouring tiles: all column tiles of an activation row-tile run on the same XCD.
#include "relax_internal.h"

namespace relax {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 struct did not)

constexpr int BK = 32;
constexpr int LDK = BK + 4;

struct GemmParams {
    const float* in;
    const float* w;
    const float* bias;
    const float* residual;
    float* out;
    int M, N, Kpad;
    int H, W, Cin, cin_log2, Ho, Wo, KW, stride, pad, ntaps;
    int act;
    int tiles_n, ntiles;
};

__device__ inline int xcd_remap(int b, int nwg) {
    // blocks b, b+8, ... share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles.
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

__device__ inline float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

template <int BM, int BN, int WM, int WN, bool TAPS>
__global__ __launch_bounds__(256) void conv_gemm_f32(const GemmParams p) {
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_LOADS = BM / 32;
    constexpr int B_LOADS = BN / 32;
    constexpr int STAGE = (BM + BN) * LDK;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3;  // 0..31: row within a 32-row staging pass
    const int c4 = tid & 7;     // which float4 of the 32-float K step

    const int tile = xcd_remap(blockIdx.x, p.ntiles);
    const int m0 = (tile / p.tiles_n) * BM;
    const int n0 = (tile % p.tiles_n) * BN;

    // per-thread row descriptors of the A operand (fixed for the whole K loop)
    int64_t a_base[A_LOADS];
    int a_iy[A_LOADS], a_ix[A_LOADS];
    bool a_ok[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + i * 32 + lrow;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (TAPS || p.stride != 1) {
            const int img = mm / (p.Ho * p.Wo);
            const int rem = mm - img * (p.Ho * p.Wo);
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_base[i] = (((int64_t)img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.Cin;
        } else {
            a_iy[i] = 0;
            a_ix[i] = 0;
            a_base[i] = (int64_t)mm * p.Cin;
        }
    }
    const float* wrow0 = p.w + (int64_t)(n0 + lrow) * p.Kpad + c4 * 4;
    const int64_t wpass = (int64_t)32 * p.Kpad;  // 32 weight rows per staging pass

    f32x4 ra[A_LOADS], rb[B_LOADS];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define RELAX_LOAD_TILE(k0_)                                                                          \
    {                                                                                                 \
        const int k_ = (k0_) + c4 * 4;                                                                \
        if (TAPS) {                                                                                   \
            const int tap_ = k_ >> p.cin_log2;                                                        \
            const int c_ = k_ & (p.Cin - 1);                                                          \
            const int dy_ = tap_ / p.KW;                                                              \
            const int dx_ = tap_ - dy_ * p.KW;                                                        \
            const int koff_ = (dy_ * p.W + dx_) * p.Cin + c_;                                         \
            const bool tap_ok_ = tap_ < p.ntaps;                                                      \
            _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                     \
                const bool ok_ = a_ok[i] && tap_ok_ && (unsigned)(a_iy[i] + dy_) < (unsigned)p.H &&   \
                                 (unsigned)(a_ix[i] + dx_) < (unsigned)p.W;                           \
                f32x4 v_ = zero4;                                                                     \
                if (ok_) v_ = *reinterpret_cast<const f32x4*>(p.in + a_base[i] + koff_);              \
                ra[i] = v_;                                                                           \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                     \
                f32x4 v_ = zero4;                                                                     \
                if (a_ok[i]) v_ = *reinterpret_cast<const f32x4*>(p.in + a_base[i] + k_);             \
                ra[i] = v_;                                                                           \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                           \
            rb[i] = *reinterpret_cast<const f32x4*>(wrow0 + i * wpass + (k0_));                       \
    }
#define RELAX_STORE_TILE(buf_)                                                                        \
    {                                                                                                 \
        float* As_ = smem + (buf_) * STAGE;                                                           \
        float* Bs_ = As_ + BM * LDK;                                                                  \
        _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                           \
            *reinterpret_cast<f32x4*>(As_ + (i * 32 + lrow) * LDK + c4 * 4) = ra[i];                  \
        _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                           \
            *reinterpret_cast<f32x4*>(Bs_ + (i * 32 + lrow) * LDK + c4 * 4) = rb[i];                  \
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.Kpad / BK;
    RELAX_LOAD_TILE(0);
    RELAX_STORE_TILE(0);
    __syncthreads();

    const int frag_off = (lane & 31) * LDK + 4 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) RELAX_LOAD_TILE((kt + 1) * BK);  // global loads in flight under the MFMAs below
        const float* As = smem + cur * STAGE + (wm * TM * 32) * LDK + frag_off;
        const float* Bs = smem + cur * STAGE + BM * LDK + (wn * TN * 32) * LDK + frag_off;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK + q * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) RELAX_STORE_TILE(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    const int64_t o = (int64_t)row * p.N + col;
                    float v = acc[i][j][r] + bias;
                    if (p.residual) v += p.residual[o];
                    p.out[o] = apply_act(v, p.act);
                }
            }
        }
    }
}

#undef RELAX_LOAD_TILE
#undef RELAX_STORE_TILE

template <int BM, int BN, int WM, int WN, bool TAPS>
static int launch_variant(relax_handle* h, GemmParams& p, hipStream_t s) {
    p.tiles_n = p.N / BN;
    p.ntiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    constexpr size_t lds = sizeof(float) * 2 * (BM + BN) * LDK;
    static bool attr_set = false;
    if (!attr_set) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_f32<BM, BN, WM, WN, TAPS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_gemm_f32<BM, BN, WM, WN, TAPS>), dim3(p.ntiles), dim3(256), lds, s, p);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

int launch_conv(relax_handle* h, const ConvDesc& d, hipStream_t s) {
    GemmParams p{};
    p.in = d.in; p.w = d.w; p.bias = d.bias; p.residual = d.residual; p.out = d.out;
    p.M = d.Nimg * d.Ho * d.Wo;
    p.N = d.Cout;
    p.Kpad = d.Kpad;
    p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Ho = d.Ho; p.Wo = d.Wo;
    p.KW = d.KW; p.stride = d.stride; p.pad = d.pad; p.ntaps = d.KH * d.KW;
    p.act = d.act;
    const bool taps = d.KH * d.KW > 1;
    RELAX_REQUIRE(h, p.M > 0 && p.N > 0 && p.Kpad > 0, "conv/gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.Kpad);
    RELAX_REQUIRE(h, p.Kpad % BK == 0, "conv/gemm: K=%d must be a multiple of %d", p.Kpad, BK);
    RELAX_REQUIRE(h, p.N % 64 == 0, "conv/gemm: N=%d must be a multiple of 64", p.N);
    RELAX_REQUIRE(h, p.Cin % 4 == 0, "conv/gemm: Cin=%d must be a multiple of 4", p.Cin);
    RELAX_REQUIRE(h, (reinterpret_cast<uintptr_t>(d.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0,
                  "conv/gemm: operands must be 16-byte aligned");
    if (taps) {
        p.cin_log2 = ilog2_exact(d.Cin);
        RELAX_REQUIRE(h, p.cin_log2 >= 2, "conv: Cin=%d must be a power of two >= 4 for KHxKW > 1", d.Cin);
        RELAX_REQUIRE(h, p.Kpad >= d.KH * d.KW * d.Cin, "conv: Kpad too small");
    } else {
        RELAX_REQUIRE(h, d.pad == 0, "conv: 1x1 with padding is not supported");
        RELAX_REQUIRE(h, p.Kpad == d.Cin, "conv/gemm: 1x1 needs Kpad == Cin");
    }
    const double flops = d.flops > 0 ? d.flops : 2.0 * p.M * (double)p.N * (double)(d.KH * d.KW * d.Cin);
    int span;
    RELAX_TRY(prof_begin(h, s, 0, flops, &span));
    // tile choice: the biggest tile that still gives every CU (256) at least ~2 workgroups
    const long tiles128 = (p.N % 128 == 0) ? (long)((p.M + 127) / 128) * (p.N / 128) : 0;
    const long tiles128x64 = (long)((p.M + 127) / 128) * (p.N / 64);
    int rc;
    if (tiles128 >= 512) {
        rc = taps ? launch_variant<128, 128, 2, 2, true>(h, p, s) : launch_variant<128, 128, 2, 2, false>(h, p, s);
    } else if (tiles128x64 >= 512) {
        rc = taps ? launch_variant<128, 64, 2, 2, true>(h, p, s) : launch_variant<128, 64, 2, 2, false>(h, p, s);
    } else {
        rc = taps ? launch_variant<64, 64, 2, 2, true>(h, p, s) : launch_variant<64, 64, 2, 2, false>(h, p, s);
    }
    RELAX_TRY(rc);
    RELAX_TRY(prof_end(h, s, span));
    return RELAX_OK;
}

}  // namespace relax

using namespace relax;

extern "C" {

int relax_op_gemm(relax_handle* h, const float* A, const float* W, const float* bias, const float* residual,
                  float* out, int M, int N, int K, int act, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, A && W && out, "relax_op_gemm: NULL operand");
    RELAX_REQUIRE(h, act >= 0 && act <= 2, "relax_op_gemm: act=%d", act);
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    return launch_gemm(h, A, W, bias, residual, out, M, N, K, act, static_cast<hipStream_t>(stream));
}

int relax_op_conv2d_nhwc(relax_handle* h, const float* in, const float* w, const float* bias, const float* residual,
                         float* out, int Nimg, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                         int act, relax_stream stream) {
    if (!h) return RELAX_ERR_INVALID;
    RELAX_REQUIRE(h, in && w && out, "relax_op_conv2d_nhwc: NULL operand");
    RELAX_REQUIRE(h, Nimg > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                  "relax_op_conv2d_nhwc: bad geometry");
    RELAX_REQUIRE(h, act >= 0 && act <= 2, "relax_op_conv2d_nhwc: act=%d", act);
    RELAX_HIP_CHECK(h, hipSetDevice(h->device));
    ConvDesc d{};
    d.in = in; d.Nimg = Nimg; d.H = H; d.W = W; d.Cin = Cin;
    d.Ho = (H + 2 * pad - KH) / stride + 1;
    d.Wo = (W + 2 * pad - KW) / stride + 1;
    d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad;
    d.w = w; d.Cout = Cout;
    d.Kpad = ((KH * KW * Cin + BK - 1) / BK) * BK;
    d.bias = bias; d.residual = residual; d.out = out; d.act = act;
    return launch_conv(h, d, static_cast<hipStream_t>(stream));
}

}  // extern "C"
