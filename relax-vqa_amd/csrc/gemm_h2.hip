// "f16x2": the fp32-grade contraction kernel on the fp16 matrix cores of gfx950 (gemm_precision 3; plain GEMMs with N % 256 == 0:
// every GEMM of the ViT).
//
//   out[m, n] = act( (sum_k A'[m, k] * W'[n, k]) * colscale[n] * rowscale[m] + bias[n] + residual[m, n] )
//
// Every fp32 operand is held as TWO fp16 numbers of a power-of-two multiple of itself (csrc/h2.h): A'[m,k] = A[m,k] * sa = ah + al,
// W'[n,k] = W[n,k] * t_n = bh + bl, 22 significant bits each.  v_mfma_f32_16x16x32_f16 contracts 32 deep; a chunk holds 16 k - so the
// 32-deep K of ONE instruction takes BOTH planes of the chunk, and
//        A[hi|lo] . B[lo|lo] = ah bl + al bl            A[hi|lo] . B[hi|hi] = ah bh + al bh              (smallest first)
// are ALL FOUR partial products in two instructions per 16 x 16 outputs and 16 k (bf16x6: six products in three), every fp16 x fp16
// product exact in the fp32 accumulator, which is rounded once per instruction.  The sum is multiplied by 2^-(t_n + e_a) in the epilogue
// (a power of two: exact).  4 bytes per operand value instead of 6: a third fewer matrix-pipe cycles, DMA pieces, LDS bytes and
// fragment reads per fp32 product than gemm_x6.hip.  Measured (tools/micro/mfma_f16x2.hip, profiles/r05_mfma_f16x2.txt): the error
// against the exact product of the fp32 operands is BELOW the fp32 FMA chain's at every K (8.9e-8 / 2.8e-7 / 5.5e-7 at K = 32 / 768 /
// 3072; chain 9.0e-8 / 4.4e-7 / 8.8e-7; bf16x6 6.7e-8 / 3.7e-7 / 7.6e-7), fp16 subnormal operands are kept by the instruction,
// and the LDS-fed loop runs 1.46 x the algorithmic rate of the bf16x6 loop at the same executed PFLOP/s (the part is power-limited).
//
// Range.  fp16 has 5 exponent bits: a value is representable with its 22 bits for 2^-3 <= |x * s| < 65504, with an absolute error of
// 2^-25 / s below that window - harmless, the errors of a dot product add absolutely and the row's large terms carry 2^-22 of their
// own size - and NOT AT ALL above it.  So the scale of every tensor comes from a bound that holds for every input:
//   * weights: per output row, from the row's own maximum, at load time (max |W'| in [2^14, 2^15));
//   * ViT activations: one static power of two per tensor (host_logic.h: LayerNorm outputs are bounded by sqrt(dim - 1) |gamma| +
//     |beta| whatever the input; the outputs of a Linear behind a LayerNorm by Cauchy-Schwarz; attention outputs are convex
//     combinations of V rows; |GELU(x)| <= |x|; patch pixels are in [0, 1]).  The bounds are 5 - 8 binades above the typical value, the
//     window is 19 binades wide.  No data-dependent scale: a row's bits do not depend on the batch, and nothing can overflow;
//   * operator-level entry points (relax_op_gemm): per row of A from that row's maximum (to_h2_rows), again batch-invariant.
// Non-finite inputs stay non-finite (inf - inf in the lo plane gives NaN), as on the other paths.
//
// Structure: the M16 loop of gemm_x6.hip on 64-byte rows.  256 x 256 tile on 8 waves (2 x 4, 128 x 64 per wave, one workgroup per CU),
// BK = 16, NSTG LDS stages of 512 rows x 64 B = 32 KB filled by LDS-DMA (4 pieces of 1 KiB per wave and step, issued one per MFMA
// group: the texture-address FIFO argument of gemm_x6.hip), counted s_waitcnt vmcnt, one raw s_barrier per step; B (2 forms x 4
// fragments) double-buffered in registers across steps, A (ONE form) streamed one 16-row fragment at a time: 16 ds_read_b128 and 64
// MFMAs per wave and step.  LDS image of a stage: the hi planes of the 512 rows as [row][32 B], then their lo planes the same way, 16 KB
// further.  With 32-byte rows the 16-lane groups of a ds_read_b128 touch every bank once for all three read patterns with NO
// permutation (8 rows span the 64 banks; a group holds rows {0-3, 12-15} at one 16-byte half and rows {4-11} at the other), the
// lo fragment of a row is the hi fragment's address + 16384 (an immediate offset: one address register per operand), and a DMA
// piece is 32 rows of one plane: the hi and the lo piece of the same rows share their per-lane source offset (the plane is +32 B in
// the scalar offset).
// Epilogue, tail split-K, XCD remap and tile grouping as gemm_x6.hip.
#include "relax_internal.h"
#include "host_logic.h"
#include "h2.h"
#include "gelu.h"

// Phase-stamp hooks of the diagnostic build: empty in the product (tools/abl/gemm_h2_stamps.hip defines them and #includes this file).
#ifndef H3_EP_FETCH_AHEAD
#define H3_EP_FETCH_AHEAD 0
#endif
#ifndef H2_STAMP
#define H2_STAMP(i_)
#define H2_STAMPS_BEFORE_LAUNCH(h_, p_, units_)
#define H2_STAMPS_AFTER_LAUNCH(h_, p_, units_, s_)
#endif
// hooks of the diagnostic build "h3b2b" (tools/build_ablations.sh; WRONG results, timing only - what a conv2 -> conv3 fusion of layer3 / layer4
// could return at most): the 1x1 launches with a residual (the conv3s) never fetch their activation rows, the 3x3 launches never store planes
#ifndef H3_ABL_NO_A
#define H3_ABL_NO_A(taps_, perimg_, p_) false
#define H3_ABL_NO_OUT_H2(taps_, p_) false
#endif
#ifndef H3_KSTEP
#define H3_KSTEP(k_) (k_)   // the K step a DMA piece of a plain gemm_h3 reads (the diagnostic build can wrap it so that every piece hits L2)
#endif

namespace relax {

typedef float h2k_f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2k_f16x8 __attribute__((ext_vector_type(8)));

// ---- conversions ---------------------------------------------------------------------------------------------------------
// fp32 [rows][K] (row stride ld floats) -> h2 [rows][K*4 B] with value * scale * row_scale[row] (either factor optional); one
// thread per 8 values
__global__ __launch_bounds__(256) void to_h2_kernel(const float* __restrict__ x, int64_t ld, char* __restrict__ y, int K, int64_t total8,
                                                    float scale, const float* __restrict__ row_scale, int rows_per_scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int k8 = K >> 3;
    const int64_t row = i / k8;
    const int k = (int)(i - row * k8) * 8;
    const float* s = x + row * ld + k;
    const float f = row_scale ? scale * row_scale[row / rows_per_scale] : scale;
    store_h2_x8(y + row * (int64_t)K * 4, k, *reinterpret_cast<const h2k_f32x4*>(s), *reinterpret_cast<const h2k_f32x4*>(s + 4), f);
}

// the same with the scale taken from the row itself: one wave per row, scale = the power of two that puts the row's maximum into
// [2^14, 2^15); inv_scale[row] = 1 / scale goes to the GEMM epilogue (operator-level entry points: no static bound on A there)
__global__ __launch_bounds__(256) void to_h2_rows_kernel(const float* __restrict__ x, int64_t ld, char* __restrict__ y, int K, int rows,
                                                         float* __restrict__ inv_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;   // wave-uniform
    const float* s = x + (int64_t)row * ld;
    float m = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const h2k_f32x4 a = *reinterpret_cast<const h2k_f32x4*>(s + k), b = *reinterpret_cast<const h2k_f32x4*>(s + k + 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float sc = h2_scale_for(m);
    if (lane == 0) inv_scale[row] = 1.f / sc;
    char* d = y + (int64_t)row * K * 4;
    for (int k = lane * 8; k < K; k += 512)
        store_h2_x8(d, k, *reinterpret_cast<const h2k_f32x4*>(s + k), *reinterpret_cast<const h2k_f32x4*>(s + k + 4), sc);
}

int launch_to_h2(relax_handle* h, const float* x, int64_t ld, void* y, int64_t rows, int K, float scale, const float* row_scale,
                 hipStream_t s, int rows_per_scale) {
    RELAX_REQUIRE(h, K % 16 == 0 && ld % 4 == 0 && rows > 0 && rows_per_scale >= 1, "to_h2: K=%d must be a multiple of 16 (ld %lld)", K, (long long)ld);
    const int64_t total8 = rows * (K / 8);
    hipLaunchKernelGGL(to_h2_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, x, ld, static_cast<char*>(y), K, total8,
                       scale, row_scale, rows_per_scale);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- per-image scales -------------------------------------------------------------------------------------------------------
// ResNet-50's activations have no input-independent bound (50 ReLU layers: any norm bound compounds), so a tensor that travels as
// fp16 planes gets one power-of-two scale per IMAGE, from a bound on that image's values that is known BEFORE the tensor is
// written: Hoelder's inequality on the producing convolution,
//        max |out| <= max_n sum_k |W[n,k]| * max |in|  (+ the second source's term) + max |bias| + max |residual|,
// with the maxima of the inputs MEASURED (the producer of each input atomicMax-es the bits of its non-negative outputs into the
// image's slot; integer max: order-free, so the value does not depend on the batch) - a bound that is loose by the ratio of the L1
// bound to the real maximum of ONE layer (5 - 7 binades of fp16's 19-binade window) and never compounds.  Everything on the device,
// no host synchronisation: one tiny launch per convolution turns the maxima into scales.
//   scale[i] = the power of two that puts (la * amax_a[i] + lb * amax_b[i] + amax_r[i] + bmax) into [2^14, 2^15),  inv[i] = 1 / scale[i]
__global__ __launch_bounds__(256) void h2_image_scales_kernel(const unsigned* __restrict__ amax_a, float la, const unsigned* __restrict__ amax_b,
                                                              float lb, const unsigned* __restrict__ amax_r, float bmax,
                                                              float* __restrict__ scale, float* __restrict__ inv, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float b = la * __uint_as_float(amax_a[i]) + bmax;
    if (amax_b) b += lb * __uint_as_float(amax_b[i]);
    if (amax_r) b += __uint_as_float(amax_r[i]);
    const float sc = h2_scale_for(b * 1.0001f);   // (the sum above is rounded: a hair of margin; the window has a factor 2 to spare)
    scale[i] = sc;
    inv[i] = 1.f / sc;
}

int launch_h2_image_scales(relax_handle* h, const unsigned* amax_a, float la, const unsigned* amax_b, float lb, const unsigned* amax_r,
                           float bmax, float* scale, float* inv, int n, hipStream_t s) {
    RELAX_REQUIRE(h, amax_a && scale && inv && n > 0, "h2_image_scales: bad arguments");
    hipLaunchKernelGGL(h2_image_scales_kernel, dim3((n + 255) / 256), dim3(256), 0, s, amax_a, la, amax_b, lb, amax_r, bmax, scale, inv, n);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// max |x| over each image's per_image values (grid: blocks x images; operator-level conversions)
__global__ __launch_bounds__(256) void image_absmax_kernel(const float* __restrict__ x, int64_t per_image, unsigned* __restrict__ amax) {
    const float* p = x + (int64_t)blockIdx.y * per_image;
    float m = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < per_image; i += (int64_t)gridDim.x * 1024) {
        const h2k_f32x4 v = *reinterpret_cast<const h2k_f32x4*>(p + i);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(amax + blockIdx.y, __float_as_uint(m));
}

int launch_image_absmax(relax_handle* h, const float* x, int64_t per_image, int n_images, unsigned* amax, hipStream_t s) {
    RELAX_REQUIRE(h, x && amax && n_images > 0 && per_image > 0 && per_image % 4 == 0, "image_absmax: bad arguments");
    RELAX_HIP_CHECK(h, hipMemsetAsync(amax, 0, sizeof(unsigned) * (size_t)n_images, s));
    const int blocks = (int)((per_image / 4 + 255) / 256 < 64 ? (per_image / 4 + 255) / 256 : 64);
    hipLaunchKernelGGL(image_absmax_kernel, dim3(blocks, n_images), dim3(256), 0, s, x, per_image, amax);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_to_h2_rows(relax_handle* h, const float* x, int64_t ld, void* y, int rows, int K, float* inv_scale, hipStream_t s) {
    RELAX_REQUIRE(h, K % 16 == 0 && ld % 4 == 0 && rows > 0, "to_h2_rows: K=%d must be a multiple of 16 (ld %lld)", K, (long long)ld);
    hipLaunchKernelGGL(to_h2_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ld, static_cast<char*>(y), K, rows, inv_scale);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

// ---- the kernel ------------------------------------------------------------------------------------------------------
struct H2Params {
    const char* a;           // h2 activations [M][K*4 B]
    const char* w;           // h2 weights [N][K*4 B]
    const float* colscale;   // [N]: 2^-(t_n + e_a), the inverse of (weight row scale x static activation scale)
    const float* rowscale;   // [M]: the inverse of a per-row activation scale (operator level), or null
    const float* bias;
    const float* residual;   // fp32 [M][N] or null
    float* out;              // fp32 [M][N] or null
    char* out_h2;            // h2 [M][N*4 B] or null, values scaled by out_scale
    float out_scale;
    float* partial;          // split-K partial tiles
    int M, N, K;
    int act;
    int no_split;
    int tiles_m, tiles_n, ntiles, group_m;
    int full_tiles, nsplit;
    // gemm_h3 as a convolution (ResNet-50 layer3 / layer4): the rows of A are the output pixels of NHWC images [Nimg*H*W][Cin*4 B],
    // taken at (oy, ox) * stride - pad (+ the tap's (dy, dx) for KH*KW > 1, implicit GEMM); weight rows k = (dy*KW + dx)*Cin + c
    int pixels, H, W, Cin, Ho, Wo, KW, stride, pad;
    // per-IMAGE scales (activations without an input-independent bound: see the scale kernel below): image = row / rows_per_img
    int rows_per_img;               // Ho*Wo; 0 = no per-image tables
    const float* img_in_inv;        // [Nimg]: 1 / (scale of image i's A planes), or null
    const float* img_out_scale;     // [Nimg]: scale of image i's rows of out_h2, or null (then out_scale)
    unsigned* amax_out;             // [Nimg]: atomicMax of the bits of the (non-negative: ReLU) outputs of image i, or null
    const char* residual_h2;        // the residual as fp16 planes [M][N*4 B] (hi + lo, times img_res_inv[image]: exact), or null
    const float* img_res_inv;
    float* gap;                     // fused spatial mean, stage 1: sums of the outputs over aligned 4-row groups [M/4][N] (rows < gap_rows), or null
    int out_rows, gap_rows;         // rows below these limits get the fp32 output / the group sums
    unsigned long long* stamps;   // diagnostic builds only; null in the product
};

__device__ inline int xcd_remap_h2(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

#define H2_DMA(rsrc_, lds_off_, voff_, soff_)                                                                             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_, (__attribute__((address_space(3))) void*)(smem + (lds_off_)), 16,     \
                                             voff_, soff_, 0, 0)
[[maybe_unused]] constexpr unsigned kH2OutOfRange = 0x80000000u;   // > every num_records below: the DMA writes zeros for that lane
constexpr int64_t kH2MaxRecords = 0x7ffffff0;

constexpr int H2_BM = 256, H2_BN = 256;
constexpr int H2_STAGE = (H2_BM + H2_BN) * kH2ChunkBytes;   // 32 KB
constexpr int H2_PIECES = H2_STAGE / 1024;                  // 32
[[maybe_unused]] constexpr int H2_PPW = H2_PIECES / 8;                       // 4 per wave: 2 activation pieces, 2 weight pieces

template <int NSTG>
__global__ __launch_bounds__(512, 2) void gemm_h2(const H2Params p) {
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (no __amdgpu_buffer_rsrc_t there)
    constexpr int BM = H2_BM, BN = H2_BN, STAGE = H2_STAGE, PPW = H2_PPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // work unit -> (tile, K slice)
    int tile, kt_begin, kt_end, slice = -1, split_tile = 0;
    {
        const int nk_all = p.K >> 4;
        const int b = blockIdx.x;
        if (b < p.full_tiles) {
            tile = xcd_remap_h2(b, p.full_tiles);
            kt_begin = 0;
            kt_end = nk_all;
        } else {
            const int u = b - p.full_tiles;
            split_tile = u / p.nsplit;
            slice = u - split_tile * p.nsplit;
            tile = p.full_tiles + split_tile;
            kt_begin = (int)((int64_t)nk_all * slice / p.nsplit);
            kt_end = (int)((int64_t)nk_all * (slice + 1) / p.nsplit);
        }
    }
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int g = tile / per_group;
        const int first = g * p.group_m;
        const int gsz = p.tiles_m - first < p.group_m ? p.tiles_m - first : p.group_m;
        const int w = tile - g * per_group;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    H2_STAMP(0);

    // ---- DMA descriptors.  A stage = [hi image: 512 rows x 32 B][lo image: the same]; a piece = 1 KiB = 32 rows of one image; lane l
    // fills half l & 1 of row 32 * (piece & 15) + (l >> 1).  This wave's four pieces: hi and lo of the activation rows 32 wave .. + 31
    // (pieces wave, 16 + wave) and of the weight rows 32 wave .. + 31 (pieces 8 + wave, 24 + wave): two per-lane offsets.
    const int64_t row_bytes = (int64_t)p.K * 4;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
    {
        const int64_t left = (int64_t)(p.M - m0) * row_bytes;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a + (int64_t)m0 * row_bytes), 0,
                                                   (int)(left < kH2MaxRecords ? left : kH2MaxRecords), 0x00020000);
        rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w + (int64_t)n0 * row_bytes), 0, (int)(BN * row_bytes), 0x00020000);
    }
    unsigned voff_a, voff_w;
    {
        const int r = 32 * wave + (lane >> 1);
        voff_a = m0 + r < p.M ? (unsigned)(r * (int)row_bytes + (lane & 1) * 16) : kH2OutOfRange;
        voff_w = (unsigned)(r * (int)row_bytes + (lane & 1) * 16);
    }
    int d_kt = kt_begin;   // the next K step to issue (steps are issued in order)
    // piece j of the wave: 0 = activations hi, 1 = weights hi, 2 = activations lo, 3 = weights lo
#define H2_ISSUE_PIECE(st_, j_)                                                                                         \
    {                                                                                                                   \
        const int soff_ = d_kt * kH2ChunkBytes + ((j_) >> 1) * 32;                                                      \
        const int dst_ = (st_) * STAGE + ((j_) >> 1) * (STAGE / 2) + ((j_) & 1) * (BM * 32) + wave * 1024;               \
        if (((j_) & 1) == 0) { H2_DMA(rsrc_a, dst_, voff_a, soff_); }                                                    \
        else { H2_DMA(rsrc_w, dst_, voff_w, soff_); }                                                                    \
    }
#define H2_ISSUE(st_)                                                                                                   \
    {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) H2_ISSUE_PIECE(st_, j);                                         \
        ++d_kt;                                                                                                         \
    }

    floatx4 acc[8][4];   // 16-row A fragments x 16-column B fragments of the wave's 128 x 64
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: lane (r, g), g = the 8-k group of the 32-deep instruction: A[hi|lo] takes half g & 1 of the hi (g < 2) or
    // the lo (g >= 2) plane, B[hi|hi] / B[lo|lo] half g & 1 of the hi / lo plane
    const int r16 = lane & 15, g16 = lane >> 4;
    const int a_hl = (g16 >> 1) * (STAGE / 2) + (wm * 128 + r16) * 32 + ((g16 & 1) << 4);
    const int b_hh = (BM + wn * 64 + r16) * 32 + ((g16 & 1) << 4);
    h2k_f16x8 xb[2][4][2], ya[2];
#define H2_READ_XB(set_, sp_)                                                                                           \
    _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                                                     \
        xb[set_][y][0] = *reinterpret_cast<const h2k_f16x8*>((sp_) + b_hh + y * 16 * 32);                               \
        xb[set_][y][1] = *reinterpret_cast<const h2k_f16x8*>((sp_) + b_hh + y * 16 * 32 + STAGE / 2);                   \
    }
#define H2_READ_A(buf_, e_, sp_)                                                                                        \
    {                                                                                                                   \
        ya[buf_] = *reinterpret_cast<const h2k_f16x8*>((sp_) + a_hl + (e_) * 16 * 32);                                  \
        __builtin_amdgcn_sched_barrier(0);   /* issued HERE, a whole MFMA group ahead of its use */                     \
    }
#define H2_MFMAS(set_, buf_, e_)                                                                                        \
    {                                                                                                                   \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya[buf_], xb[set_][y][1], acc[e_][y], 0, 0, 0);         \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya[buf_], xb[set_][y][0], acc[e_][y], 0, 0, 0);         \
        __builtin_amdgcn_sched_barrier(0);   /* nothing moves across: the software pipeline below is the schedule */     \
    }
    // region k: [wait for step k+1, barrier, B forms and A fragment 0 of step k+1, MFMAs of fragment 7 of step k, then fragments 0..6
    // of step k+1 each behind the read of the next one]; xs_ = k & 1 (register set of B), st_ = k % NSTG (LDS stage of step k, which
    // every wave has left at the barrier: the pieces of step k + NSTG go into it, one per MFMA group).  In flight at the wait: the
    // steps k+1 .. k+NSTG-1; the newest (NSTG - 2) * PPW pieces of this wave may stay in flight.
#define H2_REGION(xs_, st_, has_next_, has_d_)                                                                          \
    {                                                                                                                   \
        const char* sn_ = smem + (((st_) + 1) % NSTG) * STAGE;                                                          \
        if (has_next_) {                                                                                                \
            if (has_d_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSTG - 2) * PPW) : "memory");              \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
            __builtin_amdgcn_s_barrier();                                                                               \
            H2_READ_XB((xs_) ^ 1, sn_);                                                                                 \
            H2_READ_A(0, 0, sn_);                                                                                       \
        }                                                                                                               \
        H2_MFMAS(xs_, 1, 7);                                                                                            \
        if (has_next_) {                                                                                                \
            _Pragma("unroll") for (int e = 0; e < 7; ++e) {                                                             \
                H2_READ_A((e + 1) & 1, e + 1, sn_);                                                                     \
                if ((has_d_) && e < PPW) H2_ISSUE_PIECE(st_, e);                                                        \
                H2_MFMAS((xs_) ^ 1, e & 1, e);                                                                          \
            }                                                                                                           \
            if (has_d_) ++d_kt;                                                                                         \
        }                                                                                                               \
    }

    const int nk = kt_end - kt_begin;
    // prologue: the first NSTG steps (as many as exist) are requested; step 0 has to have landed, the younger ones stay in flight
    {
        const int pre = nk < NSTG ? nk : NSTG;
        for (int i = 0; i < pre; ++i) H2_ISSUE(i);
        if (pre >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (pre == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (pre == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    H2_READ_XB(0, smem);
    H2_READ_A(0, 0, smem);
    H2_STAMP(1);
#pragma unroll
    for (int e = 0; e < 7; ++e) {
        H2_READ_A((e + 1) & 1, e + 1, smem);
        H2_MFMAS(0, e & 1, e);
    }
    {
        // has_next = step k+1 exists; has_d = step k + NSTG exists (its DMA is issued in region k); literal `true` in the steady state.
        // One period of (register set, stage) = lcm(2, NSTG) regions, spelled out so that every index is a constant.
        constexpr int PERIOD = (NSTG % 2 == 0) ? NSTG : 2 * NSTG;
        int k = 0;
        for (; k + PERIOD + NSTG <= nk; k += PERIOD) {
#pragma unroll
            for (int i = 0; i < PERIOD; ++i) H2_REGION(i & 1, i % NSTG, true, true);
        }
        for (; k < nk; k += PERIOD) {
#pragma unroll
            for (int i = 0; i < PERIOD; ++i)
                if (k + i < nk) H2_REGION(i & 1, i % NSTG, k + i + 1 < nk, k + i + NSTG < nk);
        }
    }
#undef H2_REGION
#undef H2_MFMAS
#undef H2_READ_A
#undef H2_READ_XB
#undef H2_ISSUE
#undef H2_ISSUE_PIECE
    H2_STAMP(2);
    __syncthreads();   // no DMA is in flight and every fragment is in registers: the stages become the epilogue staging

    // ---- epilogue, staged through LDS in 64-row passes (C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r); a
    // thread owns two groups of 4 columns: adjacent (8 consecutive columns = whole 16-byte plane units) when planes leave, half a
    // tile row apart for plain fp32 traffic (every store / load instruction then covers contiguous 512-byte row segments)
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int LDC = BN + 4;
    constexpr int EP_ROWS = 64;
    constexpr int C8 = BN / 8;
    constexpr int NT = 512;
    static_assert(EP_ROWS * LDC * 4 <= NSTG * STAGE, "epilogue pass must fit the staging LDS");
    constexpr int EP_STEP = NT / C8;
    constexpr int EP_ITERS = EP_ROWS / EP_STEP;
    const bool planes = p.out_h2 != nullptr;   // workgroup-uniform
    const int lcA = planes ? (tid % C8) * 8 : (tid % C8) * 4;
    const int lcB = planes ? lcA + 4 : lcA + BN / 2;
    const int lr0 = tid / C8;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias_a = zero4, bias_b = zero4, cs_a = zero4, cs_b = zero4;
    if (slice < 0) {
        cs_a = *reinterpret_cast<const f32x4*>(p.colscale + n0 + lcA);
        cs_b = *reinterpret_cast<const f32x4*>(p.colscale + n0 + lcB);
        if (p.bias) {
            bias_a = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcA);
            bias_b = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcB);
        }
    }
    const bool interior = m0 + BM <= p.M;   // workgroup-uniform
#pragma unroll
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
        if (pass > 0) __syncthreads();
        f32x4 ra[EP_ITERS], rb[EP_ITERS];
        float rs[EP_ITERS];
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int m = m0 + pass * EP_ROWS + it * EP_STEP + lr0;
            ra[it] = zero4;
            rb[it] = zero4;
            rs[it] = 1.f;
            if (slice < 0 && (interior || m < p.M)) {
                if (p.residual) {
                    const float* r = p.residual + (int64_t)m * p.N + n0;
                    ra[it] = *reinterpret_cast<const f32x4*>(r + lcA);
                    rb[it] = *reinterpret_cast<const f32x4*>(r + lcB);
                }
                if (p.rowscale) rs[it] = p.rowscale[m];
            }
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int rb0 = wm * 128 + x * 16;
            if (rb0 / EP_ROWS == pass) {
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(rb0 % EP_ROWS + 4 * g16 + r) * LDC + wn * 64 + y * 16 + r16] = acc[x][y][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int lr = it * EP_STEP + lr0;
            const int trow = pass * EP_ROWS + lr;
            const int m = m0 + trow;
            if (!interior && m >= p.M) continue;
            f32x4 va = *reinterpret_cast<const f32x4*>(stg + lr * LDC + lcA);
            f32x4 vb = *reinterpret_cast<const f32x4*>(stg + lr * LDC + lcB);
            if (slice >= 0) {   // K slice of a split tile: raw partial sums, the epilogue runs in splitk_finish_h2
                float* o = p.partial + ((int64_t)split_tile * p.nsplit + slice) * (BM * BN) + (int64_t)trow * BN;
                *reinterpret_cast<f32x4*>(o + lcA) = va;
                *reinterpret_cast<f32x4*>(o + lcB) = vb;
                continue;
            }
            // powers of two: exact.  Then (acc + bias) + residual, the same order on every path
            va = va * (cs_a * rs[it]) + bias_a;
            vb = vb * (cs_b * rs[it]) + bias_b;
            if (p.residual) {
                va += ra[it];
                vb += rb[it];
            }
            if (p.act == 1) {
                va = (f32x4){fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f)};
                vb = (f32x4){fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f)};
            } else if (p.act == 2) {
                va = gelu_erf4(va);
                vb = gelu_erf4(vb);
            }
            const int64_t o = (int64_t)m * p.N + n0;
            if (p.out) {
                *reinterpret_cast<f32x4*>(p.out + o + lcA) = va;
                *reinterpret_cast<f32x4*>(p.out + o + lcB) = vb;
            }
            if (p.out_h2) store_h2_x8(p.out_h2 + (int64_t)m * ((int64_t)p.N * 4), n0 + lcA, va, vb, p.out_scale);   // (planes: lcB = lcA + 4)
        }
    }
    H2_STAMP(3);
#endif
}

// ---- the same contraction with a 32-deep K step ("BK = 32"), and the form the product runs ----------------------------------
// One instruction contracts the SAME plane of two consecutive 16-k chunks: lanes with k group g < 2 hold the halves of chunk 2j,
// g >= 2 those of chunk 2j + 1, i.e. the fragment is 32 consecutive k of one plane.  Per 32 k and 16 x 16 outputs
//        A[lo] . B[hi]      A[hi] . B[lo]      A[hi] . B[hi]                                    (smallest first)
// THREE instructions: the product al bl (2^-22 of the full product, the size of the representation error itself) is dropped for
// K >= 256, where the measured error against fp64 is the same with and without it (tools/micro/mfma_f16x2.hip part 1, (b) against (c):
// 2.83e-7 / 2.84e-7 at K = 768, 5.47e-7 / 5.47e-7 at K = 3072 - the fp32 accumulator's own rounding is what is left - and both below the
// fp32 FMA chain's 4.4e-7 / 8.8e-7); FOUR = true adds A[lo] . B[lo] in front (short K: at K = 32 the three-product error, 9.7e-8, would
// sit above the chain's 9.0e-8).  48 MFMAs per wave and 16 k (v1 above: 64, bf16x6: 96), 12 ds_read_b128 (16, 32), half the barriers.
// LDS: two stages of [hi image: 512 rows x 64 B][lo image: the same] = 64 KB; unit u (8 k) of row r sits in slot u ^ f((r >> 2) & 3),
// f = {0, 3, 2, 1}: conflict-free ds_read_b128 (a 16-lane group holds rows {0-3, 12-15} at one parity of g and {4-11} at the other);
// the permutation is applied on the SOURCE offset of the DMA.  The global format keeps its 16-k chunks ([16 hi][16 lo]): unit u of the
// 32-k step = chunk u >> 1, half u & 1.  Eight DMA pieces per wave and step, one in front of each MFMA group of the region; the pieces of
// step k+2 go into the stage step k has just left and have the rest of the region to land (two stages: nothing stays in flight across a
// barrier).
constexpr int H3_STAGE = (H2_BM + H2_BN) * 128;   // 64 KB
constexpr int H3_NSTG = 2;

// the maximum over the 16 lanes of a DPP row, in every lane of the row (quad swaps, then the two mirrors)
__device__ inline float dpp_row_max(float v) {
#define H3_DPP_MAX(CTRL_) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL_, 0xf, 0xf, true)))
    H3_DPP_MAX(0xB1);    // quad_perm [1, 0, 3, 2]
    H3_DPP_MAX(0x4E);    // quad_perm [2, 3, 0, 1]
    H3_DPP_MAX(0x141);   // row_half_mirror
    H3_DPP_MAX(0x140);   // row_mirror
#undef H3_DPP_MAX
    return v;
}

// PERIMG: the per-image features of the convolution launches (per-image scales, plane residual, fused means, maxima, out_rows): the
// plain GEMMs of the ViT keep the lean epilogue
template <bool FOUR, bool TAPS = false, bool PERIMG = false>
__global__ __launch_bounds__(512, 2) void gemm_h3(const H2Params p) {
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (no __amdgpu_buffer_rsrc_t there)
    constexpr int BM = H2_BM, BN = H2_BN, STAGE = H3_STAGE, NSTG = H3_NSTG, IMG = STAGE / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // work unit -> (tile, K slice)
    int tile, kt_begin, kt_end, slice = -1, split_tile = 0;
    {
        const int nk_all = p.K >> 5;   // 32-k steps
        const int b = blockIdx.x;
        if (b < p.full_tiles) {
            tile = xcd_remap_h2(b, p.full_tiles);
            kt_begin = 0;
            kt_end = nk_all;
        } else {
            const int u = b - p.full_tiles;
            split_tile = u / p.nsplit;
            slice = u - split_tile * p.nsplit;
            tile = p.full_tiles + split_tile;
            kt_begin = (int)((int64_t)nk_all * slice / p.nsplit);
            kt_end = (int)((int64_t)nk_all * (slice + 1) / p.nsplit);
        }
    }
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int g = tile / per_group;
        const int first = g * p.group_m;
        const int gsz = p.tiles_m - first < p.group_m ? p.tiles_m - first : p.group_m;
        const int w = tile - g * per_group;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;
#ifdef H3_STAGGER   // (diagnostic, tools/build_ablations.sh h3stagger:<cycles>) the workgroups of the FIRST round start (blockIdx & 3) x H3_STAGGER cycles
    if (PERIMG && !TAPS && blockIdx.x < 256 && (blockIdx.x & 3)) {   // apart, so that the CUs do not reach their epilogues together
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), wait = (unsigned long long)(blockIdx.x & 3) * (H3_STAGGER);
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
    H2_STAMP(0);

    // ---- DMA descriptors.  A piece = 1 KiB = 16 rows x 64 B of one image; the images hold 32 pieces each (activation rows: 0 .. 15,
    // weight rows: 16 .. 31).  Piece j of this wave: image j >> 2 (hi, lo), piece wave + 8 (j & 3) of it; lane l fills slot l & 3 of row
    // 16 * piece + (l >> 2), which holds unit slot ^ f(row) = chunk (unit >> 1), half (unit & 1) of the step.  The hi and the lo piece
    // of the same rows share the per-lane offset (the plane is + 32 B in the scalar offset): four offsets.
    // Convolutions (p.pixels): an activation row is an output pixel; its offset is the input pixel it reads at tap (0, 0), the step's
    // tap adds (dy W + dx) pixels, a tap outside the image (bit mask per piece) or a row past M an out-of-range offset: the DMA
    // writes zeros.  K order: 32-channel groups outermost, the KH*KW taps of a group innermost (they re-read neighbouring pixels: L2).
    const int64_t row_bytes = (int64_t)p.K * 4;                               // weight rows
    const int64_t arow_bytes = p.pixels ? (int64_t)p.Cin * 4 : row_bytes;     // activation rows / pixels
    const int img0 = p.pixels ? m0 / (p.Ho * p.Wo) : 0;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
    {
        const int64_t base = p.pixels ? (int64_t)img0 * p.H * p.W * arow_bytes : (int64_t)m0 * arow_bytes;
        const int64_t total = p.pixels ? (int64_t)(p.M / (p.Ho * p.Wo)) * p.H * p.W * arow_bytes : (int64_t)p.M * arow_bytes;
        const int64_t left = total - base;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a + base), 0, (int)(left < kH2MaxRecords ? left : kH2MaxRecords), 0x00020000);
        rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w + (int64_t)n0 * row_bytes), 0, (int)(BN * row_bytes), 0x00020000);
    }
    unsigned voff[4];
    [[maybe_unused]] unsigned a_taps[2] = {0u, 0u};   // TAPS: bit t set = tap t of this piece's pixel lies inside the image (KH*KW <= 32)
    const int ntaps = TAPS ? p.K / p.Cin : 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int trow = (wave + 8 * q) * 16 + (lane >> 2);           // 0 .. 511
        const int unit = (lane & 3) ^ ((4 - ((trow >> 2) & 3)) & 3);
        const int off = (unit >> 1) * kH2ChunkBytes + (unit & 1) * 16;
        if (q >= 2) {
            voff[q] = (unsigned)((trow - BM) * (int)row_bytes + off);
        } else if (m0 + trow >= p.M || H3_ABL_NO_A(TAPS, PERIMG, p)) {
            voff[q] = kH2OutOfRange;
        } else if (p.pixels) {
            const int m = m0 + trow;
            const int img = m / (p.Ho * p.Wo);
            const int rem = m - img * (p.Ho * p.Wo);
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const int iy = oy * p.stride - p.pad, ix = ox * p.stride - p.pad;
            if (TAPS) {
                for (int t = 0, dy = 0, dx = 0; t < ntaps; ++t) {
                    if ((unsigned)(iy + dy) < (unsigned)p.H && (unsigned)(ix + dx) < (unsigned)p.W) a_taps[q] |= 1u << t;
                    if (++dx == p.KW) { dx = 0; ++dy; }
                }
            }
            voff[q] = (unsigned)((((img - img0) * p.H + iy) * p.W + ix) * (int)arow_bytes + off);
        } else {
            voff[q] = (unsigned)(trow * (int)arow_bytes + off);
        }
    }
    // the next 32-k step to issue (steps are issued in order): plain = a byte offset; taps = (32-channel group, dy, dx)
    int d_kt = kt_begin;
    [[maybe_unused]] int d_cc = 0, d_tap = 0, d_dy = 0, d_dx = 0;
    [[maybe_unused]] const int cin_chunks = p.Cin >> 4;
    if (TAPS) {
        d_cc = kt_begin / ntaps;
        d_tap = kt_begin - d_cc * ntaps;
        d_dy = d_tap / p.KW;
        d_dx = d_tap - d_dy * p.KW;
    }
#define H3_ISSUE_PIECE(st_, j_)                                                                                         \
    {                                                                                                                   \
        const int dst_ = (st_) * STAGE + ((j_) >> 2) * IMG + (wave + 8 * ((j_) & 3)) * 1024;                             \
        if (((j_) & 3) < 2) {                                                                                           \
            if (TAPS) {                                                                                                 \
                const bool ok_ = (a_taps[(j_) & 1] >> d_tap) & 1u;                                                      \
                const int tapoff_ = (d_dy * p.W + d_dx) * (int)arow_bytes + d_cc * (2 * kH2ChunkBytes) + ((j_) >> 2) * 32; \
                H2_DMA(rsrc_a, dst_, ok_ ? voff[(j_) & 3] + (unsigned)tapoff_ : kH2OutOfRange, 0);                      \
            } else {                                                                                                    \
                H2_DMA(rsrc_a, dst_, voff[(j_) & 3], H3_KSTEP(d_kt) * (2 * kH2ChunkBytes) + ((j_) >> 2) * 32);          \
            }                                                                                                           \
        } else {                                                                                                        \
            const int wk_ = TAPS ? (d_tap * cin_chunks + 2 * d_cc) * kH2ChunkBytes : H3_KSTEP(d_kt) * (2 * kH2ChunkBytes); \
            H2_DMA(rsrc_w, dst_, voff[(j_) & 3], wk_ + ((j_) >> 2) * 32);                                                \
        }                                                                                                               \
    }
#define H3_ISSUE_ADVANCE()                                                                                              \
    {                                                                                                                   \
        ++d_kt;                                                                                                         \
        if (TAPS) {                                                                                                     \
            ++d_tap;                                                                                                    \
            ++d_dx;                                                                                                     \
            const bool wx_ = d_dx == p.KW;                                                                              \
            d_dx = wx_ ? 0 : d_dx;                                                                                      \
            d_dy += wx_ ? 1 : 0;                                                                                        \
            const bool wt_ = d_tap == ntaps;                                                                            \
            d_tap = wt_ ? 0 : d_tap;                                                                                    \
            d_dy = wt_ ? 0 : d_dy;                                                                                      \
            d_cc += wt_ ? 1 : 0;                                                                                        \
        }                                                                                                               \
    }

    floatx4 acc[8][4];   // 16-row A fragments x 16-column B fragments of the wave's 128 x 64
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: lane (r, g) reads unit g (k = 8g .. 8g+7 of the 32-k step) of its row, hi image; lo image: + IMG
    const int r16 = lane & 15, g16 = lane >> 4;
    const int f16 = (4 - ((r16 >> 2) & 3)) & 3;
    const int a_h = (wm * 128 + r16) * 64 + ((g16 ^ f16) << 4);
    const int b_h = (BM + wn * 64 + r16) * 64 + ((g16 ^ f16) << 4);
    h2k_f16x8 xb[2][4][2], ya[2][2];   // [set][fragment][hi, lo], [buffer][hi, lo]
#define H3_READ_XB(set_, sp_)                                                                                           \
    _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                                                     \
        xb[set_][y][0] = *reinterpret_cast<const h2k_f16x8*>((sp_) + b_h + y * 16 * 64);                                \
        xb[set_][y][1] = *reinterpret_cast<const h2k_f16x8*>((sp_) + b_h + y * 16 * 64 + IMG);                          \
    }
#define H3_READ_A(buf_, e_, sp_)                                                                                        \
    {                                                                                                                   \
        ya[buf_][0] = *reinterpret_cast<const h2k_f16x8*>((sp_) + a_h + (e_) * 16 * 64);                                \
        ya[buf_][1] = *reinterpret_cast<const h2k_f16x8*>((sp_) + a_h + (e_) * 16 * 64 + IMG);                          \
        __builtin_amdgcn_sched_barrier(0);   /* issued HERE, a whole MFMA group ahead of their use */                   \
    }
#define H3_MFMAS(set_, buf_, e_)                                                                                        \
    {                                                                                                                   \
        if (FOUR) {                                                                                                     \
            _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                               \
                acc[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya[buf_][1], xb[set_][y][1], acc[e_][y], 0, 0, 0);  \
        }                                                                                                               \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya[buf_][1], xb[set_][y][0], acc[e_][y], 0, 0, 0);      \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya[buf_][0], xb[set_][y][1], acc[e_][y], 0, 0, 0);      \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                                   \
            acc[e_][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya[buf_][0], xb[set_][y][0], acc[e_][y], 0, 0, 0);      \
        __builtin_amdgcn_sched_barrier(0);   /* nothing moves across: the software pipeline below is the schedule */     \
    }
    // region k: [wait for step k+1 (every piece of this wave: two stages), barrier, B forms and A fragment 0 of step k+1, piece 0 of step
    // k+2 into the stage step k has left, MFMAs of fragment 7 of step k, then fragments 0..6 of step k+1 each behind the read of the next
    // one and one more piece]; xs_ = k & 1 = the register set of B = the LDS stage of step k.
    // the order the eight pieces of a step are issued in: activation and weight pieces interleaved (A, A, W, W of the hi planes, then of
    // the lo planes).  Measured, cycles per 16 k of the qkv / fc2 loops: this order 1767 / 1956, all four activation pieces first 1836 /
    // 2070, all four weight pieces first 1943 / 2116 (profiles/r05_h3_group_m_and_dma_placement.txt)
#define H3_ORDER(i_) (i_)
#define H3_REGION(xs_, has_next_, has_d_)                                                                               \
    {                                                                                                                   \
        const char* sn_ = smem + ((xs_) ^ 1) * STAGE;                                                                   \
        if (has_next_) {                                                                                                \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
            __builtin_amdgcn_s_barrier();                                                                               \
            H3_READ_XB((xs_) ^ 1, sn_);                                                                                 \
            H3_READ_A(0, 0, sn_);                                                                                       \
            if (has_d_) H3_ISSUE_PIECE(xs_, H3_ORDER(0));                                                               \
        }                                                                                                               \
        H3_MFMAS(xs_, 1, 7);                                                                                            \
        if (has_next_) {                                                                                                \
            _Pragma("unroll") for (int e = 0; e < 7; ++e) {                                                             \
                H3_READ_A((e + 1) & 1, e + 1, sn_);                                                                     \
                if (has_d_) H3_ISSUE_PIECE(xs_, H3_ORDER(e + 1));                                                       \
                H3_MFMAS((xs_) ^ 1, e & 1, e);                                                                          \
            }                                                                                                           \
            if (has_d_) H3_ISSUE_ADVANCE();                                                                             \
        }                                                                                                               \
    }

    const int nk = kt_end - kt_begin;
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) H3_ISSUE_PIECE(0, j);
        H3_ISSUE_ADVANCE();
        if (nk > 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) H3_ISSUE_PIECE(1, j);
            H3_ISSUE_ADVANCE();
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // step 0 has landed; step 1 stays in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    H3_READ_XB(0, smem);
    H3_READ_A(0, 0, smem);
    H2_STAMP(1);
#pragma unroll
    for (int e = 0; e < 7; ++e) {
        H3_READ_A((e + 1) & 1, e + 1, smem);
        H3_MFMAS(0, e & 1, e);
    }
    {
        // has_next = step k+1 exists; has_d = step k+2 exists (its DMA is issued in region k); literal `true` in the steady state
        int k = 0;
        for (; k + 3 < nk; k += 2) {
            H3_REGION(0, true, true);
            H3_REGION(1, true, true);
        }
        for (; k < nk; k += 2) {
            H3_REGION(0, k + 1 < nk, k + 2 < nk);
            if (k + 1 < nk) H3_REGION(1, k + 2 < nk, k + 3 < nk);
        }
    }
#undef H3_REGION
#undef H3_ORDER
#undef H3_MFMAS
#undef H3_READ_A
#undef H3_READ_XB
#undef H3_ISSUE_PIECE
#undef H3_ISSUE_ADVANCE
    H2_STAMP(2);
    __syncthreads();   // no DMA is in flight and every fragment is in registers: the stages become the epilogue staging
    H2_STAMP(7);

    // ---- epilogue, staged through LDS in 64-row passes (C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r); a
    // thread owns two groups of 4 columns: adjacent (8 consecutive columns = whole 16-byte plane units) when planes leave, half a
    // tile row apart for plain fp32 traffic (every store / load instruction then covers contiguous 512-byte row segments)
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int LDC = BN + 4;
    constexpr int EP_ROWS = 64;
    constexpr int C8 = BN / 8;
    constexpr int NT = 512;
    static_assert(EP_ROWS * LDC * 4 <= NSTG * STAGE, "epilogue pass must fit the staging LDS");
    constexpr int EP_STEP = NT / C8;
    constexpr int EP_ITERS = EP_ROWS / EP_STEP;
    const bool planes = p.out_h2 != nullptr || (PERIMG && p.residual_h2 != nullptr);   // workgroup-uniform
    const int lcA = planes ? (tid % C8) * 8 : (tid % C8) * 4;
    const int lcB = planes ? lcA + 4 : lcA + BN / 2;
    const int lr0 = tid / C8;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // Column scales and bias: UNCONDITIONAL loads (a K slice loads and ignores them; no bias = a buffer resource of zero bytes, whose loads
    // answer zeros).  Loaded under `if (p.bias)` the value is a phi with the zero default, which the compiler resolved by a register copy
    // BEHIND the load - an s_waitcnt on the load's latency at the head of every tile's epilogue, long before the first use.
    const f32x4 cs_a = *reinterpret_cast<const f32x4*>(p.colscale + n0 + lcA);
    const f32x4 cs_b = *reinterpret_cast<const f32x4*>(p.colscale + n0 + lcB);
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    const f32x4 bias_a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (n0 + lcA) * 4, 0, 0));
    const f32x4 bias_b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (n0 + lcB) * 4, 0, 0));
    const bool interior = m0 + BM <= p.M;   // workgroup-uniform
    // per-image maxima of this tile, behind the staging area (which takes the first EP_ROWS * LDC * 4 = 66.5 KB of the stages)
    unsigned* simg = reinterpret_cast<unsigned*>(smem + 96 * 1024);
    // row -> image without a division per row: the tile's first image and its row offset inside that image are workgroup-uniform, the
    // quotient of the small remainder comes from a float product and one correction step either way (exact: the operands are < 2^24)
    const int rpi = p.rows_per_img;
    const int img_first = rpi > 0 ? m0 / rpi : 0;
    const int row_first = m0 - img_first * rpi;
    const float rpi_inv = rpi > 0 ? 1.f / (float)rpi : 0.f;
    auto img_of_row = [&](int trow) -> int {
        const int x = row_first + trow;
        int q = (int)((float)x * rpi_inv);
        q -= (q * rpi > x) ? 1 : 0;
        q += ((q + 1) * rpi <= x) ? 1 : 0;
        return img_first + q;
    };
    const bool lds_amax = PERIMG && p.amax_out && slice < 0 && p.rows_per_img >= 18;   // at most 16 images under the tile's 256 rows
    if (lds_amax && tid < 16) simg[tid] = 0u;   // (ordered before the first use by the barrier of the first pass)
    // Addresses: the fp32 matrices and the plane matrices have the same row pitch (4 N bytes) and a 256-column tile starts 4 n0 bytes into
    // the row in both, so ONE workgroup-uniform byte offset places the tile in all of them (64-bit, scalar registers); a thread adds a
    // 32-bit offset - its row inside the tile, its 8 columns as fp32 (ofs_f, second group ofs_fb further) or as plane units (ofs_p) - and
    // the row step of a pass is uniform again: loads and stores take the scalar-base form, one VALU add per row instead of a 64-bit
    // multiply per access
    const int64_t tile_ofs = ((int64_t)m0 * p.N + n0) * 4;
    const unsigned row_pitch = (unsigned)p.N * 4u;
    const unsigned ofs_f = (unsigned)lr0 * row_pitch + (unsigned)lcA * 4u;
    const unsigned ofs_p = (unsigned)lr0 * row_pitch + (unsigned)h2_offset(lcA);
    const unsigned ofs_fb = (unsigned)(lcB - lcA) * 4u;
    // The residual - fp32 rows or, convolution form, planes - through ONE buffer resource that starts at the tile and ends with the matrix
    // (rows beyond M answer zeros; no residual, or a K slice: a resource of zero bytes): the loads are unconditional.  Loaded under
    // `if (p.residual)` the rows were phis with their zero defaults, and the compiler resolved one of them by a register copy BEHIND the
    // load: an s_waitcnt on the first load of every row before the next row's loads went out - the fetch of a pass was four latencies long.
    const bool res_planes = PERIMG && p.residual_h2 != nullptr;   // workgroup-uniform
    const char* res_ptr = res_planes ? p.residual_h2 : reinterpret_cast<const char*>(p.residual);
    const int64_t res_left = (int64_t)p.M * p.N * 4 - tile_ofs;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(res_ptr) + (res_ptr ? tile_ofs : 0), 0, (res_ptr && slice < 0) ? (int)(res_left < kH2MaxRecords ? res_left : kH2MaxRecords) : 0, 0x00020000);
    const unsigned ofs_r = res_planes ? ofs_p : ofs_f, ofs_rb = res_planes ? 32u : ofs_fb;
    char* tile_out = reinterpret_cast<char*>(p.out) + tile_ofs;
    char* tile_out_h2 = p.out_h2 + tile_ofs;
    // What a pass needs from memory for its rows - residual, scales - is fetched by H3_EP_FETCH at the head of the pass (a plane residual
    // stays RAW in ra / rb - hi plane, lo plane - and is decoded where it is consumed).  H3_EP_FETCH_AHEAD = 1 (convolution form) fetches
    // pass 0 here and pass p + 1 from inside the row loop of pass p, into the registers the row has just consumed, so that the fetch would
    // be in flight under the rest of the pass: measured twice, never faster - with per-access 64-bit addresses the form needed 256
    // registers and spilled 7 pointers, and every scratch reload in the row loop is an s_waitcnt vmcnt(0) that waits for the fetch it
    // follows (epilogue of layer4's conv3 98.7 k -> 117.8 k cycles per tile); with the scalar-base addresses above 4 spills remain, outside
    // the row loop, and a ResNet-50 pass takes the same 36.5 ms either way (that epilogue 76 k -> 86 k by the stamps): the wait for the
    // residual rows is not what bounds these epilogues.  Off (tools/build_ablations.sh h3ahead, tools/r06_h3ahead.sh).
    f32x4 ra[EP_ITERS], rb[EP_ITERS];
    float rs[EP_ITERS], os[EP_ITERS], rri[EP_ITERS];
    int im[EP_ITERS];
#define H3_EP_FETCH(P_, IT_)                                                                                                     \
    {                                                                                                                            \
        const int trow_ = (P_) * EP_ROWS + (IT_) * EP_STEP + lr0;                                                                \
        const int m_ = m0 + trow_;                                                                                               \
        const unsigned row_ofs_ = (unsigned)(((P_) * EP_ROWS + (IT_) * EP_STEP) * row_pitch);                                    \
        ra[IT_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, row_ofs_ + ofs_r, 0, 0));              \
        rb[IT_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, row_ofs_ + ofs_r + ofs_rb, 0, 0));     \
        rs[IT_] = 1.f;                                                                                                           \
        os[IT_] = p.out_scale;                                                                                                   \
        rri[IT_] = 1.f;                                                                                                          \
        im[IT_] = 0;                                                                                                             \
        if (slice < 0 && (interior || m_ < p.M)) {                                                                               \
            if (p.rowscale) rs[IT_] = p.rowscale[m_];                                                                            \
            if (PERIMG && rpi > 0) { /* per-image scales: the row's image */                                                     \
                im[IT_] = img_of_row(trow_);                                                                                     \
                if (p.img_in_inv) rs[IT_] = p.img_in_inv[im[IT_]];                                                               \
                if (p.img_out_scale) os[IT_] = p.img_out_scale[im[IT_]];                                                         \
            }                                                                                                                    \
            if (res_planes) rri[IT_] = p.img_res_inv[im[IT_]]; /* (planes: ra = the hi unit, rb = the lo unit, lcB = lcA + 4) */ \
        }                                                                                                                        \
    }
    constexpr bool EP_AHEAD = PERIMG && H3_EP_FETCH_AHEAD;
    if constexpr (EP_AHEAD) {
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) H3_EP_FETCH(0, it);
    }
#pragma unroll
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
        if (pass > 0) __syncthreads();
        if constexpr (!EP_AHEAD) {
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) H3_EP_FETCH(pass, it);
        }
        if (pass == 0) H2_STAMP(4);   // (diagnostic builds: the phases of the first pass)
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int rb0 = wm * 128 + x * 16;
            if (rb0 / EP_ROWS == pass) {
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(rb0 % EP_ROWS + 4 * g16 + r) * LDC + wn * 64 + y * 16 + r16] = acc[x][y][r];
            }
        }
        __syncthreads();
        if (pass == 0) H2_STAMP(5);
        if constexpr (!EP_AHEAD) {
            // Everything the pass fetched has to be in its registers HERE, before the first store of the row loop: the empty statement reads
            // and redefines the registers, so the compiler waits for the loads now (they went out before the staging and the barrier) and
            // for nothing later.  Left alone it waits in front of each row for that row's loads with a count that must hold on every
            // path through the loop's (uniform) branches - the one without stores - and on the real path that count also covers the
            // stores of the rows before: every row waited for the write acknowledgements of the previous one.
#pragma unroll
            for (int it = 0; it < EP_ITERS; ++it) {
                asm volatile("" : "+v"(ra[it]), "+v"(rb[it]), "+v"(rs[it]));
                if constexpr (PERIMG) asm volatile("" : "+v"(os[it]), "+v"(rri[it]));
            }
        }
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int lr = it * EP_STEP + lr0;
            const int trow = pass * EP_ROWS + lr;
            const int m = m0 + trow;
            // this row's operands leave the fetch registers (a plane residual is decoded: (hi + lo) / scale is the stored 22-bit value,
            // exactly)
            f32x4 xa = ra[it], xb = rb[it];
            const float rs_it = rs[it], os_it = os[it];
            const int im_it = im[it];
            if (res_planes) {
                const h2_u32x4 hi = __builtin_bit_cast(h2_u32x4, xa), lo = __builtin_bit_cast(h2_u32x4, xb);
                const float ri = rri[it];
#define H2_LO16(u_) ((float)__builtin_bit_cast(_Float16, (unsigned short)((u_) & 0xffffu)))
#define H2_HI16(u_) ((float)__builtin_bit_cast(_Float16, (unsigned short)((u_) >> 16)))
                xa = (f32x4){H2_LO16(hi.x) + H2_LO16(lo.x), H2_HI16(hi.x) + H2_HI16(lo.x), H2_LO16(hi.y) + H2_LO16(lo.y),
                             H2_HI16(hi.y) + H2_HI16(lo.y)} * ri;
                xb = (f32x4){H2_LO16(hi.z) + H2_LO16(lo.z), H2_HI16(hi.z) + H2_HI16(lo.z), H2_LO16(hi.w) + H2_LO16(lo.w),
                             H2_HI16(hi.w) + H2_HI16(lo.w)} * ri;
#undef H2_LO16
#undef H2_HI16
            }
            if constexpr (EP_AHEAD) {
                if (pass + 1 < BM / EP_ROWS) H3_EP_FETCH(pass + 1, it);
            }
            if (!interior && m >= p.M) continue;
            f32x4 va = *reinterpret_cast<const f32x4*>(stg + lr * LDC + lcA);
            f32x4 vb = *reinterpret_cast<const f32x4*>(stg + lr * LDC + lcB);
            if (slice >= 0) {   // K slice of a split tile: raw partial sums, the epilogue runs in splitk_finish_h2
                float* o = p.partial + ((int64_t)split_tile * p.nsplit + slice) * (BM * BN) + (int64_t)trow * BN;
                *reinterpret_cast<f32x4*>(o + lcA) = va;
                *reinterpret_cast<f32x4*>(o + lcB) = vb;
                continue;
            }
            // powers of two: exact.  Then (acc + bias) + residual, the same order on every path
            va = va * (cs_a * rs_it) + bias_a;
            vb = vb * (cs_b * rs_it) + bias_b;
            if (res_ptr) {
                va += xa;
                vb += xb;
            }
            if (p.act == 1) {
                va = (f32x4){fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f)};
                vb = (f32x4){fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f)};
            } else if (p.act == 2) {
                va = gelu_erf4(va);
                vb = gelu_erf4(vb);
            }
            const unsigned row_ofs = (unsigned)((pass * EP_ROWS + it * EP_STEP) * row_pitch);
            if (p.out && (!PERIMG || m < p.out_rows)) {
                char* o = tile_out + (size_t)(unsigned)(row_ofs + ofs_f);
                *reinterpret_cast<f32x4*>(o) = va;
                *reinterpret_cast<f32x4*>(o + ofs_fb) = vb;
            }
            if (PERIMG && p.gap) {   // the finished values go back to the staging rows for the group sums below
                *reinterpret_cast<f32x4*>(stg + lr * LDC + lcA) = va;
                *reinterpret_cast<f32x4*>(stg + lr * LDC + lcB) = vb;
            }
            if (p.out_h2 && !H3_ABL_NO_OUT_H2(TAPS, p)) {   // (planes: lcB = lcA + 4)
                h2_u32x4 hi, lo;
                split2_x8(va * os_it, vb * os_it, hi, lo);
                char* o = tile_out_h2 + (size_t)(unsigned)(row_ofs + ofs_p);
                *reinterpret_cast<h2_u32x4*>(o) = hi;
                *reinterpret_cast<h2_u32x4*>(o + 32) = lo;
            }
            if (PERIMG && p.amax_out) {
                // the largest output of this row segment (outputs are >= 0: these launches end in a ReLU) into the image's slot: integer
                // max of the bits = float max, order-free, so the maximum - and every scale derived from it - is the same whatever batch
                // the image travels in.  The 32 lanes that share the row are two DPP rows: four DPP steps leave the maximum of 16 lanes
                // in each of them (no LDS round trips), and the first lane of either row posts it.  Collected per tile in LDS first (a
                // tile spans a few images): one global atomic per image and tile instead of one per row segment
                float mx = fmaxf(fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w)), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
                mx = dpp_row_max(mx);
                if ((tid & 15) == 0) {
                    if (lds_amax) atomicMax(simg + (im_it - img_first), __float_as_uint(mx));
                    else atomicMax(p.amax_out + im_it, __float_as_uint(mx));
                }
            }
        }
        if (pass == 0) H2_STAMP(6);
        if (PERIMG && p.gap && slice < 0) {
            // fused spatial mean, stage 1: sums over the aligned 4-row groups of this pass, rows added in order.  Images start at multiples
            // of 4 rows (the launcher checks Ho*Wo % 4 == 0): a group never spans two images and the grouping - hence every bit - does
            // not depend on where in the batch an image sits.  gap_groups_finish (layers.hip) adds an image's group sums in order.
            __syncthreads();
            for (int e = tid; e < (EP_ROWS / 4) * BN; e += NT) {
                const int g = e / BN, col = e - g * BN;
                const int mg = m0 + pass * EP_ROWS + g * 4;
                if (mg < p.gap_rows) {
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += stg[(g * 4 + r) * LDC + col];
                    p.gap[(int64_t)(mg >> 2) * p.N + n0 + col] = t;
                }
            }
        }
    }
#undef H3_EP_FETCH
    if (lds_amax) {
        __syncthreads();
        if (tid < 16 && simg[tid] != 0u) atomicMax(p.amax_out + img_first + tid, simg[tid]);
    }
    H2_STAMP(3);
#endif
}

// Sums the K slices of the split tiles in slice order (deterministic) and applies the epilogue.
__global__ __launch_bounds__(256) void splitk_finish_h2(const H2Params p) {
    constexpr int BM = H2_BM, BN = H2_BN;
    const int split_tile = blockIdx.y;
    const int tile = p.full_tiles + split_tile;
    int tm, tn;
    {
        const int per_group = p.group_m * p.tiles_n;
        const int g = tile / per_group;
        const int first = g * p.group_m;
        const int gsz = p.tiles_m - first < p.group_m ? p.tiles_m - first : p.group_m;
        const int w = tile - g * per_group;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    const int e8 = blockIdx.x * 256 + threadIdx.x;  // 8-column group index inside the tile (the grid covers the tile exactly)
    const int lr = e8 / (BN / 8), lc = (e8 % (BN / 8)) * 8;
    const int m = m0 + lr;
    // per-image maxima of this block's 8 rows (at most 8 images), collected in LDS: one global atomic per image and block
    __shared__ unsigned s_mx[8];
    const int img_b0 = p.rows_per_img > 0 ? (m0 + blockIdx.x * 8) / p.rows_per_img : 0;
    if (threadIdx.x < 8) s_mx[threadIdx.x] = 0u;
    __syncthreads();
    if (m < p.M) {
    const float* pt = p.partial + (int64_t)split_tile * p.nsplit * (BM * BN) + lr * BN + lc;
    f32x4 va = *reinterpret_cast<const f32x4*>(pt), vb = *reinterpret_cast<const f32x4*>(pt + 4);
    for (int k = 1; k < p.nsplit; ++k) {
        va += *reinterpret_cast<const f32x4*>(pt + (int64_t)k * (BM * BN));
        vb += *reinterpret_cast<const f32x4*>(pt + (int64_t)k * (BM * BN) + 4);
    }
    const int64_t o = (int64_t)m * p.N + n0 + lc;
    float rs = p.rowscale ? p.rowscale[m] : 1.f, os = p.out_scale;
    const int img = p.rows_per_img > 0 ? m / p.rows_per_img : 0;
    if (p.rows_per_img > 0) {
        if (p.img_in_inv) rs = p.img_in_inv[img];
        if (p.img_out_scale) os = p.img_out_scale[img];
    }
    va = va * (*reinterpret_cast<const f32x4*>(p.colscale + n0 + lc) * rs);
    vb = vb * (*reinterpret_cast<const f32x4*>(p.colscale + n0 + lc + 4) * rs);
    if (p.bias) {
        va += *reinterpret_cast<const f32x4*>(p.bias + n0 + lc);
        vb += *reinterpret_cast<const f32x4*>(p.bias + n0 + lc + 4);
    }
    if (p.residual) {
        va += *reinterpret_cast<const f32x4*>(p.residual + o);
        vb += *reinterpret_cast<const f32x4*>(p.residual + o + 4);
    }
    if (p.act == 1) {
        va = (f32x4){fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f)};
        vb = (f32x4){fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f)};
    } else if (p.act == 2) {
        va = gelu_erf4(va);
        vb = gelu_erf4(vb);
    }
    if (p.out && m < p.out_rows) {
        *reinterpret_cast<f32x4*>(p.out + o) = va;
        *reinterpret_cast<f32x4*>(p.out + o + 4) = vb;
    }
    if (p.out_h2) store_h2_x8(p.out_h2 + (int64_t)m * ((int64_t)p.N * 4), n0 + lc, va, vb, os);
    if (p.amax_out) {
        float mx = fmaxf(fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w)), fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)));
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 31) == 0) atomicMax(&s_mx[(img - img_b0) & 7], __float_as_uint(mx));   // (8 rows: at most 8 images)
    }
    }   // m < p.M
    if (p.amax_out) {
        __syncthreads();
        if (threadIdx.x < 8 && s_mx[threadIdx.x] != 0u) atomicMax(p.amax_out + img_b0 + threadIdx.x, s_mx[threadIdx.x]);
    }
}

// FORM 0: gemm_h2<3>, 1: gemm_h2<4> (16-k steps, four products), 2: gemm_h3<false> (32-k steps, three products), 3: gemm_h3<true> (four),
// 4: gemm_h3<false, true, true> (three products, implicit-GEMM taps, per-image epilogue), 5: gemm_h3<false, false, true> (1x1 convolutions)
template <int FORM>
static int launch_h2_variant(relax_handle* h, H2Params& p, hipStream_t s) {
    constexpr int BM = H2_BM, BN = H2_BN;
    constexpr bool BK32 = FORM >= 2;
    p.tiles_n = p.N / BN;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.ntiles = p.tiles_m * p.tiles_n;
    p.group_m = h->gemm.group_m;
    p.partial = nullptr;
    // Tail split-K: the last, partial round of tiles is cut along K (cost model: host_logic.cpp, shared with the other kernels)
    // (splitk_finish_h2 knows neither the fused group sums nor a plane residual: those launches run unsplit)
    const host::TailSplit ts = host::choose_tail_split(p.ntiles, 256, BK32 ? p.K / 32 : p.K / 16, BK32 ? 4 : 8,
                                                       h->gemm.split_k && !p.no_split && !p.gap && !p.residual_h2);
    p.full_tiles = ts.full_tiles;
    p.nsplit = ts.nsplit;
    if (p.nsplit > 1) {
        const size_t need = sizeof(float) * (size_t)(p.ntiles - p.full_tiles) * p.nsplit * BM * BN;
        RELAX_TRY(ensure_buf(h, h->splitk_ws, need < (size_t)(64 << 20) ? (size_t)(64 << 20) : need));
        p.partial = static_cast<float*>(h->splitk_ws.p);
    }
    constexpr size_t lds = BK32 ? (size_t)H3_NSTG * H3_STAGE : (size_t)(FORM == 1 ? 4 : 3) * H2_STAGE;
    auto kernel = [] {
        if constexpr (FORM == 0) return &gemm_h2<3>;
        else if constexpr (FORM == 1) return &gemm_h2<4>;
        else if constexpr (FORM == 2) return &gemm_h3<false>;
        else if constexpr (FORM == 3) return &gemm_h3<true>;
        else if constexpr (FORM == 4) return &gemm_h3<false, true, true>;
        else return &gemm_h3<false, false, true>;
    }();
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[h->device]) {
        RELAX_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[h->device] = true;
    }
    const int units = p.full_tiles + (p.ntiles - p.full_tiles) * p.nsplit;
    H2_STAMPS_BEFORE_LAUNCH(h, p, units);
    hipLaunchKernelGGL(kernel, dim3(units), dim3(512), lds, s, p);
    H2_STAMPS_AFTER_LAUNCH(h, p, units, s);
    if (p.nsplit > 1)
        hipLaunchKernelGGL(splitk_finish_h2, dim3(BM * BN / 8 / 256, p.ntiles - p.full_tiles), dim3(256), 0, s, p);
    RELAX_HIP_CHECK(h, hipGetLastError());
    return RELAX_OK;
}

int launch_gemm_h2(relax_handle* h, const GemmDescH2& d, hipStream_t s) {
    H2Params p{};
    p.a = static_cast<const char*>(d.a);
    p.w = static_cast<const char*>(d.w);
    p.colscale = d.colscale; p.rowscale = d.rowscale;
    p.bias = d.bias; p.residual = d.residual; p.out = d.out; p.out_h2 = static_cast<char*>(d.out_h2); p.out_scale = d.out_scale;
    p.M = d.M; p.N = d.N; p.K = d.K; p.act = d.act; p.no_split = d.no_split;
    const bool conv = d.pixels != 0;
    const bool taps = conv && d.KH * d.KW > 1;
    if (conv) {
        p.pixels = 1; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Ho = d.Ho; p.Wo = d.Wo; p.KW = d.KW; p.stride = d.stride; p.pad = d.pad;
        RELAX_REQUIRE(h, d.Nimg > 0 && d.M == d.Nimg * d.Ho * d.Wo && d.K == d.KH * d.KW * d.Cin && d.Cin % 32 == 0 && d.KH * d.KW <= 32 &&
                             (taps || d.pad == 0) && d.stride >= 1 && (int64_t)d.H * d.W * d.Cin * 4 * 4 < kH2MaxRecords,
                      "f16x2 conv: bad geometry (Cin=%d must be a multiple of 32, at most 32 taps, M = Nimg*Ho*Wo, K = KH*KW*Cin)", d.Cin);
    }
    p.rows_per_img = d.rows_per_img; p.img_in_inv = d.img_in_inv; p.img_out_scale = d.img_out_scale; p.amax_out = d.amax_out;
    p.residual_h2 = static_cast<const char*>(d.residual_h2); p.img_res_inv = d.img_res_inv; p.gap = d.gap_groups;
    p.out_rows = d.out_rows > 0 && d.out_rows < d.M ? d.out_rows : d.M;
    p.gap_rows = d.gap_rows > 0 && d.gap_rows < d.M ? d.gap_rows : d.M;
    RELAX_REQUIRE(h, !(d.residual && d.residual_h2) && (!d.residual_h2 || (d.img_res_inv && d.rows_per_img > 0)), "f16x2 gemm: bad residual");
    RELAX_REQUIRE(h, !d.gap_groups || (d.rows_per_img > 0 && d.rows_per_img % 4 == 0), "f16x2 conv: the fused spatial mean needs Ho*Wo %% 4 == 0");
    RELAX_REQUIRE(h, d.rows_per_img > 0 || (!d.img_in_inv && !d.img_out_scale && !d.amax_out), "f16x2 gemm: per-image tables need rows_per_img");
    RELAX_REQUIRE(h, conv || (d.rows_per_img == 0 && !d.residual_h2 && !d.gap_groups && d.out_rows == 0),
                  "f16x2 gemm: per-image scales, plane residuals, fused means and out_rows belong to the convolution form");
    RELAX_REQUIRE(h, p.M > 0 && p.N > 0 && p.K > 0, "f16x2 gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    RELAX_REQUIRE(h, p.K % 16 == 0 && p.N % 256 == 0, "f16x2 gemm: K=%d must be a multiple of 16 and N=%d of 256", p.K, p.N);
    RELAX_REQUIRE(h, (int64_t)p.K * 4 * 256 < kH2MaxRecords, "f16x2 gemm: K=%d too large", p.K);
    RELAX_REQUIRE(h, d.a && d.w && d.colscale && (d.out || d.out_h2 || d.gap_groups), "f16x2 gemm: missing operand / no output requested");
    // (the running maximum is an integer atomicMax on float BITS: ordered only for non-negative values, i.e. behind a ReLU - as gemm_x6.hip requires)
    RELAX_REQUIRE(h, !d.amax_out || d.act == 1, "f16x2 gemm: amax_out needs act == 1 (ReLU): the bit-ordered maximum is valid for non-negative outputs only");
    RELAX_REQUIRE(h, !d.out_h2 || d.img_out_scale || (d.out_scale > 0.f && d.out_scale < 3.0e38f), "f16x2 gemm: the plane output needs its scale");
    auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    RELAX_REQUIRE(h, aligned16(d.a) && aligned16(d.w) && aligned16(d.colscale) && aligned16(d.bias) && aligned16(d.residual) && aligned16(d.out) &&
                         aligned16(d.out_h2),
                  "f16x2 gemm: every operand pointer must be 16-byte aligned");
    const double flops = 2.0 * p.M * (double)p.N * (double)p.K;
    // algorithmic HBM bytes: operands as planes (4 B per value), outputs (+ residual), each touched once
    const double bytes = 4.0 * ((conv ? (double)d.Nimg * d.H * d.W * d.Cin : (double)p.M * p.K) + (double)p.N * p.K) +
                         (double)p.M * p.N * ((d.out ? 4.0 : 0.0) + (d.out_h2 ? 4.0 : 0.0) + (d.residual ? 4.0 : 0.0));
    int span;
    RELAX_TRY(prof_begin(h, s, conv ? 5 : 6, flops, &span, bytes));   // (kind 6: the plain GEMMs - the ViT - alone: the dominant kernel of the headline)
    // "h2_form": 1 (default) = 32-k steps with three products for K >= 256 and four below (and 16-k steps for K % 32 != 0);
    // 0 = 16-k steps, four products ("h2_stages" LDS stages); 2 = 32-k steps, four products at every K
    int rc;
    if (conv) {
        RELAX_REQUIRE(h, p.K % 32 == 0, "f16x2 conv: K=%d must be a multiple of 32", p.K);
        rc = taps ? launch_h2_variant<4>(h, p, s) : launch_h2_variant<5>(h, p, s);
    } else if (h->gemm.h2_form == 0 || p.K % 32 != 0) rc = h->gemm.h2_stages == 4 ? launch_h2_variant<1>(h, p, s) : launch_h2_variant<0>(h, p, s);
    else if (h->gemm.h2_form == 2 || p.K < 256) rc = launch_h2_variant<3>(h, p, s);
    else rc = launch_h2_variant<2>(h, p, s);
    if (rc != RELAX_OK) { prof_abort(h, span); return rc; }
    RELAX_TRY(prof_end(h, s, span));
    return RELAX_OK;
}

}  // namespace relax
